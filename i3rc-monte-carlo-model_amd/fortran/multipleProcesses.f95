! Fortran-95 shell of the MI355X photon-tracing integrator -- process layer.
! Public interface of the reference's module MultipleProcesses (Code/multipleProcesses_nompi.f95:15-99,
! Code/multipleProcesses_mpi.f95:17-131): MasterProc, initializeProcesses, synchronizeProcesses,
! finalizeProcesses, sumAcrossProcesses (real scalar and rank 1-4 arrays).
!
! This build is the one-process-per-node variant: the process drives its GPU through the C ABI and the
! sums are identities, exactly like the reference's _nompi file.  Multi-GPU runs shard photons over
! ranks and all-reduce the packed tally buffer with RCCL (bench.py / multigpu.py); see DESIGN.md (e).
module MultipleProcesses
  implicit none
  private
  logical, save :: MasterProc = .true.
  integer, save :: processCount = 1, processRank = 0

  interface sumAcrossProcesses
    module procedure sumScalar, sumRank1, sumRank2, sumRank3, sumRank4
  end interface sumAcrossProcesses

  public :: MasterProc, initializeProcesses, synchronizeProcesses, finalizeProcesses, sumAcrossProcesses
contains
  subroutine initializeProcesses(numProcs, thisProcNum)
    integer, intent(out) :: numProcs, thisProcNum
    processCount = 1
    processRank  = 0
    MasterProc   = .true.
    numProcs     = processCount
    thisProcNum  = processRank
  end subroutine initializeProcesses

  subroutine synchronizeProcesses
  end subroutine synchronizeProcesses

  subroutine finalizeProcesses
  end subroutine finalizeProcesses

  function sumScalar(x) result(total)
    real, intent(in) :: x
    real             :: total
    total = x
  end function sumScalar

  function sumRank1(x) result(total)
    real, dimension(:), intent(in) :: x
    real, dimension(size(x))       :: total
    total = x
  end function sumRank1

  function sumRank2(x) result(total)
    real, dimension(:, :), intent(in)       :: x
    real, dimension(size(x, 1), size(x, 2)) :: total
    total = x
  end function sumRank2

  function sumRank3(x) result(total)
    real, dimension(:, :, :), intent(in)                :: x
    real, dimension(size(x, 1), size(x, 2), size(x, 3)) :: total
    total = x
  end function sumRank3

  function sumRank4(x) result(total)
    real, dimension(:, :, :, :), intent(in)                         :: x
    real, dimension(size(x, 1), size(x, 2), size(x, 3), size(x, 4)) :: total
    total = x
  end function sumRank4
end module MultipleProcesses
