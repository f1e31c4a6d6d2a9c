! Self test of module MultipleProcesses: run one copy per rank (RANK / WORLD_SIZE / MASTER_PORT in the environment,
! I3RC_COMM_BACKEND=shm for a CPU-only run).  Every rank contributes rank-dependent values; sums are checked exactly.
program commSelfTest
  use MultipleProcesses
  implicit none
  integer :: numProcs, thisProc, i, j, k, l
  real    :: s, v1(5), v2(3, 2), v3(2, 3, 2), v4(2, 2, 2, 3), big(3000000)
  real    :: want
  real(8) :: d1(700001)   ! (an odd length beyond one staging slot of the shm backend: 2^19 float64 values)
  logical :: ok

  call initializeProcesses(numProcs, thisProc)
  want = numProcs * (numProcs + 1) / 2.          ! sum over ranks of (rank + 1)
  ok = .true.
  s = sumAcrossProcesses(real(thisProc + 1));                       ok = ok .and. s == want
  v1 = (/ (real(i * (thisProc + 1)), i = 1, 5) /)
  v1 = sumAcrossProcesses(v1);                                      ok = ok .and. all(v1 == (/ (real(i) * want, i = 1, 5) /))
  forall(i = 1:3, j = 1:2) v2(i, j) = (10 * i + j) * (thisProc + 1)
  v2 = sumAcrossProcesses(v2)
  forall(i = 1:3, j = 1:2) v2(i, j) = v2(i, j) - (10 * i + j) * want
  ok = ok .and. all(v2 == 0.)
  forall(i = 1:2, j = 1:3, k = 1:2) v3(i, j, k) = (100 * i + 10 * j + k) * (thisProc + 1)
  v3 = sumAcrossProcesses(v3)
  forall(i = 1:2, j = 1:3, k = 1:2) v3(i, j, k) = v3(i, j, k) - (100 * i + 10 * j + k) * want
  ok = ok .and. all(v3 == 0.)
  forall(i = 1:2, j = 1:2, k = 1:2, l = 1:3) v4(i, j, k, l) = (i + 2 * j + 4 * k + 8 * l) * (thisProc + 1)
  v4 = sumAcrossProcesses(v4)
  forall(i = 1:2, j = 1:2, k = 1:2, l = 1:3) v4(i, j, k, l) = v4(i, j, k, l) - (i + 2 * j + 4 * k + 8 * l) * want
  ok = ok .and. all(v4 == 0.)
  big = real(thisProc + 1)                                          ! longer than one staging slot of the shm backend
  big = sumAcrossProcesses(big);                                    ok = ok .and. all(big == want)
  ! the float64 extension: values a real(4) sum would round (1 + 2^-40 per rank), exact in float64
  d1 = (1.d0 + 2.d0**(-40)) * (thisProc + 1)
  d1(700001) = 1.d15 + thisProc
  d1 = sumAcrossProcesses(d1)
  ok = ok .and. all(d1(:700000) == (1.d0 + 2.d0**(-40)) * want) .and. d1(700001) == numProcs * 1.d15 + numProcs * (numProcs - 1) / 2
  call synchronizeProcesses
  if(ok) then
    print '(a, i0, a, i0, a, l1)', "rank ", thisProc, " of ", numProcs, " sums ok master=", MasterProc
  else
    print '(a, i0, a)', "rank ", thisProc, " WRONG SUMS"
  end if
  call finalizeProcesses
  if(.not. ok) stop 1
end program commSelfTest
