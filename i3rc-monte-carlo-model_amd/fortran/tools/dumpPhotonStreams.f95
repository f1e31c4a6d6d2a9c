! Test tool: every photon-stream constructor of module monteCarloIllumination (the reference's six, Code/
! monteCarloIllumination.f95:46-50) with fixed arguments and the seed (/ 10, 1 /); prints the streams' five arrays
! as IEEE bit patterns, one photon per line, so that tests/test_fortran_shell.py can compare them BIT FOR BIT with the
! CPU oracle's restatement of the same constructors (oracle/illumination.c).  Host code only: no GPU is touched.
!   usage: dumpPhotonStreams <number of photons>
program dumpPhotonStreams
  use ErrorMessages
  use RandomNumbers
  use monteCarloIllumination
  implicit none
  type(ErrorMessage)         :: status
  type(randomNumberSequence) :: rng
  type(photonStream)         :: photons
  character(len=32)          :: arg
  integer                    :: n

  call getarg(1, arg)
  read(arg, *) n

  rng = new_RandomNumberSequence(seed = (/ 10, 1 /))
  photons = new_PhotonStream(0.6, 135., numberOfPhotons = n, randomNumbers = rng, status = status)   ! Directional
  call dump("directional", photons)

  rng = new_RandomNumberSequence(seed = (/ 10, 1 /))
  photons = new_PhotonStream(0.6, numberOfPhotons = n, randomNumbers = rng, status = status)         ! RandomAzimuth
  call dump("randomAzimuth", photons)

  rng = new_RandomNumberSequence(seed = (/ 10, 1 /))
  photons = new_PhotonStream(numberOfPhotons = n, randomNumbers = rng, status = status)              ! Flux
  call dump("flux", photons)

  photons = new_PhotonStream(0.6, 135., 0.25, 0.75, numberOfPhotons = n, status = status)            ! Spotlight
  call dump("spotlight", photons)

  rng = new_RandomNumberSequence(seed = (/ 10, 1 /))
  photons = new_PhotonStream(0.4, 0.5, 0.3, .true., numberOfPhotons = n, randomNumbers = rng, status = status)
  call dump("internalFluxUp", photons)

  rng = new_RandomNumberSequence(seed = (/ 10, 1 /))
  photons = new_PhotonStream(0.4, 0.5, 0.3, .false., deltaX = 0.1, deltaY = 0.2, &
                             numberOfPhotons = n, randomNumbers = rng, status = status)
  call dump("internalFluxDownFinite", photons)

  rng = new_RandomNumberSequence(seed = (/ 10, 1 /))
  photons = new_PhotonStream(0.4, 0.5, 0.3, -0.7, 200., deltaX = 0.1, &
                             numberOfPhotons = n, randomNumbers = rng, status = status)
  call dump("internalIntensity", photons)
  if(stateIsFailure(status)) stop 1
contains
  subroutine dump(name, stream)
    character(len=*),   intent(in   ) :: name
    type(photonStream), intent(inout) :: stream
    real    :: x, y, z, mu, phi
    integer :: i
    i = 0
    do while(morePhotonsExist(stream))
      call getNextPhoton(stream, x, y, z, mu, phi, status)
      i = i + 1
      print '(A, 1X, I0, 5(1X, Z8.8))', trim(name), i, transfer(x, 1), transfer(y, 1), transfer(z, 1), transfer(mu, 1), transfer(phi, 1)
    end do
    call finalize_PhotonStream(stream)
  end subroutine dump
end program dumpPhotonStreams
