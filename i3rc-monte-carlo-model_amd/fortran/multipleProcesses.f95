! Fortran-95 shell of the MI355X photon-tracing integrator -- process layer.
! Public interface of the reference's module MultipleProcesses (Code/multipleProcesses_nompi.f95:15-99,
! Code/multipleProcesses_mpi.f95:17-131): MasterProc, initializeProcesses, synchronizeProcesses,
! finalizeProcesses, sumAcrossProcesses (real scalar and rank 1-4 arrays; as an extension a real(8) rank-1 array).
!
! One process per GPU.  Where the reference calls MPI_REDUCE(MPI_REAL, MPI_SUM) (:57-131) this module calls
! i3rc_comm_sum_float (include/i3rc_comm.h): an RCCL all-reduce over xGMI (every rank receives the sum, a superset of
! the reference's reduce-to-root).  Ranks come from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK); a
! single process needs no launcher and behaves like the reference's _nompi variant.
module MultipleProcesses
  use, intrinsic :: iso_c_binding
  implicit none
  private
  logical, save :: MasterProc = .true.

  interface sumAcrossProcesses
    module procedure sumScalar, sumRank1, sumRank2, sumRank3, sumRank4, sumRank1Double
  end interface sumAcrossProcesses

  interface
    function i3rc_comm_init(numProcs, thisProc) bind(C, name = "i3rc_comm_init") result(rc)
      import
      integer(c_int), intent(out) :: numProcs, thisProc
      integer(c_int)              :: rc
    end function
    function i3rc_comm_local_device() bind(C, name = "i3rc_comm_local_device") result(device)
      import
      integer(c_int) :: device
    end function
    function i3rc_comm_barrier() bind(C, name = "i3rc_comm_barrier") result(rc)
      import
      integer(c_int) :: rc
    end function
    function i3rc_comm_sum_float(values, n) bind(C, name = "i3rc_comm_sum_float") result(rc)
      import
      real(c_float), intent(inout) :: values(*)
      integer(c_int64_t), value    :: n
      integer(c_int)               :: rc
    end function
    function i3rc_comm_sum_double(values, n) bind(C, name = "i3rc_comm_sum_double") result(rc)
      import
      real(c_double), intent(inout) :: values(*)
      integer(c_int64_t), value     :: n
      integer(c_int)                :: rc
    end function
    function i3rc_comm_finalize() bind(C, name = "i3rc_comm_finalize") result(rc)
      import
      integer(c_int) :: rc
    end function
    function i3rc_comm_last_error() bind(C, name = "i3rc_comm_last_error") result(text)
      import
      type(c_ptr) :: text
    end function
    function c_strlen(text) bind(C, name = "strlen") result(n)
      import
      type(c_ptr), value :: text
      integer(c_size_t)  :: n
    end function
  end interface

  public :: MasterProc, initializeProcesses, synchronizeProcesses, finalizeProcesses, sumAcrossProcesses
  public :: localDevice   ! extension: the GPU this process drives (used by new_Integrator)
contains
  subroutine initializeProcesses(numProcs, thisProcNum)
    integer, intent(out) :: numProcs, thisProcNum
    integer(c_int) :: n, r
    if(i3rc_comm_init(n, r) /= 0) then
      print *, "initializeProcesses: cannot set up the process group: " // lastError()
      stop 1
    end if
    numProcs    = n
    thisProcNum = r
    MasterProc  = r == 0
  end subroutine initializeProcesses

  integer function localDevice()
    localDevice = i3rc_comm_local_device()
  end function localDevice

  subroutine synchronizeProcesses
    if(i3rc_comm_barrier() /= 0) then
      print *, "synchronizeProcesses failed: " // lastError()
      stop 1
    end if
  end subroutine synchronizeProcesses

  subroutine finalizeProcesses
    integer :: rc
    rc = i3rc_comm_finalize()
  end subroutine finalizeProcesses

  subroutine sumInPlace(flat)
    real, dimension(:), intent(inout) :: flat
    if(size(flat) == 0) return
    if(i3rc_comm_sum_float(flat, int(size(flat), c_int64_t)) /= 0) then
      print *, "sumAcrossProcesses failed: " // lastError()
      stop 1
    end if
  end subroutine sumInPlace

  function sumScalar(x) result(total)
    real, intent(in) :: x
    real             :: total
    real :: one(1)
    one(1) = x
    call sumInPlace(one)
    total = one(1)
  end function sumScalar

  function sumRank1(x) result(total)
    real, dimension(:), intent(in) :: x
    real, dimension(size(x))       :: total
    total = x
    call sumInPlace(total)
  end function sumRank1

  function sumRank2(x) result(total)
    real, dimension(:, :), intent(in)       :: x
    real, dimension(size(x, 1), size(x, 2)) :: total
    real, dimension(size(x))                :: flat
    flat = reshape(x, (/ size(x) /))
    call sumInPlace(flat)
    total = reshape(flat, shape(x))
  end function sumRank2

  function sumRank3(x) result(total)
    real, dimension(:, :, :), intent(in)                :: x
    real, dimension(size(x, 1), size(x, 2), size(x, 3)) :: total
    real, dimension(:), allocatable                     :: flat
    allocate(flat(size(x)))
    flat = reshape(x, (/ size(x) /))
    call sumInPlace(flat)
    total = reshape(flat, shape(x))
    deallocate(flat)
  end function sumRank3

  function sumRank4(x) result(total)
    real, dimension(:, :, :, :), intent(in)                         :: x
    real, dimension(size(x, 1), size(x, 2), size(x, 3), size(x, 4)) :: total
    real, dimension(:), allocatable                                 :: flat
    allocate(flat(size(x)))
    flat = reshape(x, (/ size(x) /))
    call sumInPlace(flat)
    total = reshape(flat, shape(x))
    deallocate(flat)
  end function sumRank4

  ! An extension (the reference's MPI_REDUCE is MPI_REAL, :57-131): one float64 all-reduce of a packed buffer -- what the build's own
  ! driver sums its device-side batch moments with (monteCarloRadiativeTransfer: sumBatchMomentsAcrossProcesses), instead of
  ! rounding them to real(4) and reducing field by field.
  function sumRank1Double(x) result(total)
    real(c_double), dimension(:), intent(in) :: x
    real(c_double), dimension(size(x))       :: total
    total = x
    if(size(total) == 0) return
    if(i3rc_comm_sum_double(total, int(size(total), c_int64_t)) /= 0) then
      print *, "sumAcrossProcesses failed: " // lastError()
      stop 1
    end if
  end function sumRank1Double

  function lastError() result(message)
    character(len = 256) :: message
    character(kind = c_char), pointer :: chars(:)
    type(c_ptr) :: text
    integer     :: i, n
    message = ""
    text = i3rc_comm_last_error()
    if(.not. c_associated(text)) return
    n = min(int(c_strlen(text)), len(message))
    call c_f_pointer(text, chars, (/ n /))
    do i = 1, n
      message(i:i) = chars(i)
    end do
  end function lastError

end module MultipleProcesses
