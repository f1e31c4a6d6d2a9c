! Fortran-95 shell of the MI355X photon-tracing integrator -- sources of photons.
! Public interface of the reference's module monteCarloIllumination (Code/monteCarloIllumination.f95:46-55):
! photonStream, new_PhotonStream (six constructors), morePhotonsExist, getNextPhoton, finalize_PhotonStream.
!
! MI355X design: the Directional stream both drivers use is held LAZILY (count + direction only).  The GPU
! generates each photon's starting point from that photon's own Philox stream, so no 20-byte-per-photon array is
! built on the host or copied over PCIe.  If somebody iterates a lazy stream with getNextPhoton it is
! materialised on demand with the host generator it was created with.  All other sources produce explicit
! arrays on the host, like the reference, and are handed to the device as they are.
module monteCarloIllumination
  use ErrorMessages, only: ErrorMessage, stateIsFailure, setStateToFailure, setStateToWarning, setStateToSuccess
  use RandomNumbers, only: randomNumberSequence, getRandomReal
  implicit none
  private

  type photonStream
    private
    integer :: currentPhoton = 0
    integer :: numberOfPhotons = 0
    logical :: lazyDirectional = .false.
    real    :: solarMu = 0., solarAzimuthDegrees = 0.
    type(randomNumberSequence)  :: generator              ! state at creation, for on-demand materialisation
    real, dimension(:), pointer :: xPosition  => null(), yPosition  => null(), zPosition => null()
    real, dimension(:), pointer :: initialMu  => null(), initialPhi => null()
  end type photonStream

  interface new_PhotonStream
    module procedure streamDirectional, streamRandomAzimuth, streamFlux, streamSpotlight, &
                     streamInternalFlux, streamInternalIntensity
  end interface new_PhotonStream

  public :: photonStream
  public :: new_PhotonStream, finalize_PhotonStream, morePhotonsExist, getNextPhoton
  public :: describeStream, consumeStream, streamArrays      ! extensions used by the GPU integrator
contains
  ! ------------------------------------------------------------------------------------------------
  subroutine checkCount(n, status)
    integer,            intent(in   ) :: n
    type(ErrorMessage), intent(inout) :: status
    if(n <= 0) call setStateToFailure(status, "setIllumination: must ask for non-negative number of photons.")
  end subroutine checkCount
  subroutine checkMu(mu, status)
    real,               intent(in   ) :: mu
    type(ErrorMessage), intent(inout) :: status
    if(abs(mu) > 1. .or. abs(mu) <= tiny(mu)) call setStateToFailure(status, "setIllumination: solarMu out of bounds")
  end subroutine checkMu
  subroutine checkAzimuth(az, status)
    real,               intent(in   ) :: az
    type(ErrorMessage), intent(inout) :: status
    if(az < 0. .or. az > 360.) call setStateToFailure(status, "setIllumination: solarAzimuth out of bounds")
  end subroutine checkAzimuth
  subroutine makeRoom(photons, n)
    type(photonStream), intent(inout) :: photons
    integer,            intent(in   ) :: n
    allocate(photons%xPosition(n), photons%yPosition(n), photons%zPosition(n), photons%initialMu(n), photons%initialPhi(n))
    photons%numberOfPhotons = n
    photons%currentPhoton = 1
    photons%lazyDirectional = .false.
  end subroutine makeRoom

  ! parallel beam from (solarMu, solarAzimuth), uniformly random entry points at the top of the domain
  function streamDirectional(solarMu, solarAzimuth, numberOfPhotons, randomNumbers, status) result(photons)
    real,                       intent(in   ) :: solarMu, solarAzimuth
    integer                                   :: numberOfPhotons
    type(randomNumberSequence), intent(inout) :: randomNumbers
    type(ErrorMessage),         intent(inout) :: status
    type(photonStream)                        :: photons
    call checkCount(numberOfPhotons, status)
    call checkAzimuth(solarAzimuth, status)
    call checkMu(solarMu, status)
    if(stateIsFailure(status)) return
    photons%lazyDirectional     = .true.
    photons%numberOfPhotons     = numberOfPhotons
    photons%solarMu             = solarMu
    photons%solarAzimuthDegrees = solarAzimuth
    photons%generator           = randomNumbers
    photons%currentPhoton       = 1
    call setStateToSuccess(status)
  end function streamDirectional

  subroutine materialise(photons)
    type(photonStream), intent(inout) :: photons
    integer :: i, n, cursor
    if(.not. photons%lazyDirectional) return
    n = photons%numberOfPhotons; cursor = photons%currentPhoton
    call makeRoom(photons, n)
    do i = 1, n
      photons%xPosition(i) = getRandomReal(photons%generator)
      photons%yPosition(i) = getRandomReal(photons%generator)
    end do
    photons%zPosition(:)  = 1. - spacing(1.)
    photons%initialMu(:)  = -abs(photons%solarMu)
    photons%initialPhi(:) = photons%solarAzimuthDegrees * acos(-1.) / 180.
    photons%currentPhoton = cursor
  end subroutine materialise

  function streamRandomAzimuth(solarMu, numberOfPhotons, randomNumbers, status) result(photons)
    real,                       intent(in   ) :: solarMu
    integer                                   :: numberOfPhotons
    type(randomNumberSequence), intent(inout) :: randomNumbers
    type(ErrorMessage),         intent(inout) :: status
    type(photonStream)                        :: photons
    integer :: i
    call checkCount(numberOfPhotons, status)
    call checkMu(solarMu, status)
    if(stateIsFailure(status)) return
    call makeRoom(photons, numberOfPhotons)
    do i = 1, numberOfPhotons
      photons%xPosition(i)  = getRandomReal(randomNumbers)
      photons%yPosition(i)  = getRandomReal(randomNumbers)
      photons%initialPhi(i) = getRandomReal(randomNumbers) * 2. * acos(-1.)
    end do
    photons%zPosition(:) = 1. - spacing(1.)
    photons%initialMu(:) = -abs(solarMu)
    call setStateToSuccess(status)
  end function streamRandomAzimuth

  ! diffuse illumination: flux on the horizontal equally weighted in mu
  function streamFlux(numberOfPhotons, randomNumbers, status) result(photons)
    integer                                   :: numberOfPhotons
    type(randomNumberSequence), intent(inout) :: randomNumbers
    type(ErrorMessage),         intent(inout) :: status
    type(photonStream)                        :: photons
    integer :: i
    call checkCount(numberOfPhotons, status)
    if(stateIsFailure(status)) return
    call makeRoom(photons, numberOfPhotons)
    do i = 1, numberOfPhotons
      photons%xPosition(i)  = getRandomReal(randomNumbers)
      photons%yPosition(i)  = getRandomReal(randomNumbers)
      photons%initialMu(i)  = -sqrt(getRandomReal(randomNumbers))
      photons%initialPhi(i) = getRandomReal(randomNumbers) * 2. * acos(-1.)
    end do
    photons%zPosition(:) = 1. - spacing(1.)
    call setStateToSuccess(status)
  end function streamFlux

  function streamSpotlight(solarMu, solarAzimuth, solarX, solarY, numberOfPhotons, randomNumbers, status) result(photons)
    real,                       intent(in   ) :: solarMu, solarAzimuth, solarX, solarY
    integer                                   :: numberOfPhotons
    type(randomNumberSequence), optional, intent(inout) :: randomNumbers
    type(ErrorMessage),         intent(inout) :: status
    type(photonStream)                        :: photons
    call checkCount(numberOfPhotons, status)
    call checkAzimuth(solarAzimuth, status)
    call checkMu(solarMu, status)
    if(solarX > 1. .or. solarX <= 0. .or. solarY > 1. .or. solarY <= 0.) &
      call setStateToFailure(status, "setIllumination: x and y positions must be between 0 and 1")
    if(stateIsFailure(status)) return
    call makeRoom(photons, numberOfPhotons)
    photons%xPosition(:)  = solarX
    photons%yPosition(:)  = solarY
    photons%zPosition(:)  = 1. - spacing(1.)
    photons%initialMu(:)  = -abs(solarMu)
    photons%initialPhi(:) = solarAzimuth * acos(-1.) / 180.
    call setStateToSuccess(status)
  end function streamSpotlight

  subroutine checkDetector(x, y, z, deltaX, deltaY, status)
    real,               intent(in   ) :: x, y, z
    real, optional,     intent(in   ) :: deltaX, deltaY
    type(ErrorMessage), intent(inout) :: status
    if(x > 1. .or. x <= 0. .or. y > 1. .or. y <= 0. .or. z > 1. .or. z <= 0.) &
      call setStateToFailure(status, "setIllumination: x, y, z positions must be between 0 and 1")
    if(present(deltaX)) then
      if(x + deltaX / 2. > 1. .or. x - deltaX / 2. <= 0.) &
        call setStateToFailure(status, "setIllumination: max, min positions must be between 0 and 1")
    end if
    if(present(deltaY)) then
      if(y + deltaY / 2. > 1. .or. y - deltaY / 2. <= 0.) &
        call setStateToFailure(status, "setIllumination: max, min positions must be between 0 and 1")
    end if
  end subroutine checkDetector

  subroutine jitter(positions, width, randomNumbers)
    real, dimension(:),         intent(inout) :: positions
    real,                       intent(in   ) :: width
    type(randomNumberSequence), intent(inout) :: randomNumbers
    integer :: i
    do i = 1, size(positions)
      positions(i) = positions(i) + width * (1. - 0.5 * getRandomReal(randomNumbers))
    end do
  end subroutine jitter

  ! backward Monte Carlo: hemispheric (flux) detector inside the domain
  function streamInternalFlux(detectorX, detectorY, detectorZ, detectorPointsUp, deltaX, deltaY, &
                              numberOfPhotons, randomNumbers, status) result(photons)
    real,                       intent(in   ) :: detectorX, detectorY, detectorZ
    logical,                    intent(in   ) :: detectorPointsUp
    real,             optional, intent(in   ) :: deltaX, deltaY
    integer                                   :: numberOfPhotons
    type(randomNumberSequence), optional, intent(inout) :: randomNumbers
    type(ErrorMessage),         intent(inout) :: status
    type(photonStream)                        :: photons
    integer :: i
    call checkCount(numberOfPhotons, status)
    call checkDetector(detectorX, detectorY, detectorZ, deltaX, deltaY, status)
    if(.not. present(randomNumbers)) call setStateToFailure(status, "setIllumination: random numbers are required")
    if(detectorPointsUp .and. abs(detectorZ - 1.) < 2. * spacing(1.)) &
      call setStateToWarning(status, "setIllumination: Detector is at top of domain pointed up")
    if(.not. detectorPointsUp .and. detectorZ < 2. * tiny(0.)) &
      call setStateToWarning(status, "setIllumination: Detector is at bottom of domain pointed down")
    if(stateIsFailure(status)) return
    call makeRoom(photons, numberOfPhotons)
    photons%xPosition(:) = detectorX
    photons%yPosition(:) = detectorY
    if(detectorPointsUp) then
      photons%zPosition(:) = max(detectorZ, 2. * tiny(0.))
    else
      photons%zPosition(:) = min(detectorZ, 1. - spacing(1.))
    end if
    do i = 1, numberOfPhotons
      photons%initialMu(i)  = sqrt(getRandomReal(randomNumbers))
      photons%initialPhi(i) = getRandomReal(randomNumbers) * 2. * acos(-1.)
    end do
    if(.not. detectorPointsUp) photons%initialMu(:) = -photons%initialMu(:)
    do i = 1, numberOfPhotons             ! a horizontal start could travel for ever in an empty layer
      do while(abs(photons%initialMu(i)) <= 2. * tiny(0.))
        photons%initialMu(i) = sqrt(getRandomReal(randomNumbers))
      end do
    end do
    if(present(deltaX)) call jitter(photons%xPosition, deltaX, randomNumbers)
    if(present(deltaY)) call jitter(photons%yPosition, deltaY, randomNumbers)
    call setStateToSuccess(status)
  end function streamInternalFlux

  ! backward Monte Carlo: radiance detector inside the domain
  function streamInternalIntensity(detectorX, detectorY, detectorZ, detectorMu, detectorPhi, deltaX, deltaY, deltaTheta, &
                                   numberOfPhotons, randomNumbers, status) result(photons)
    real,               intent(in   ) :: detectorX, detectorY, detectorZ, detectorMu, detectorPhi
    real,     optional, intent(in   ) :: deltaX, deltaY, deltaTheta
    integer                           :: numberOfPhotons
    type(randomNumberSequence), optional, intent(inout) :: randomNumbers
    type(ErrorMessage), intent(inout) :: status
    type(photonStream)                :: photons
    call checkCount(numberOfPhotons, status)
    call checkDetector(detectorX, detectorY, detectorZ, deltaX, deltaY, status)
    if(detectorPhi < 0. .or. detectorPhi > 360.) call setStateToFailure(status, "setIllumination: detectorPhi out of bounds")
    if(abs(detectorMu) > 1. .or. abs(detectorMu) <= tiny(detectorMu)) &
      call setStateToFailure(status, "setIllumination: detectorMu out of bounds")
    if((present(deltaX) .or. present(deltaY)) .and. .not. present(randomNumbers)) &
      call setStateToFailure(status, "setIllumination: random numbers are required for a finite detector")
    if(present(deltaTheta)) &
      call setStateToWarning(status, "setIllumination: Finite detector angular width not yet implemented")
    if(stateIsFailure(status)) return
    call makeRoom(photons, numberOfPhotons)
    photons%xPosition(:)  = detectorX
    photons%yPosition(:)  = detectorY
    photons%initialMu(:)  = detectorMu
    photons%initialPhi(:) = detectorPhi
    if(detectorMu > tiny(detectorMu)) then
      photons%zPosition(:) = max(detectorZ, 2. * tiny(0.))
    else
      photons%zPosition(:) = min(detectorZ, 1. - spacing(1.))
    end if
    if(present(deltaX)) call jitter(photons%xPosition, deltaX, randomNumbers)
    if(present(deltaY)) call jitter(photons%yPosition, deltaY, randomNumbers)
    call setStateToSuccess(status)
  end function streamInternalIntensity

  ! ------------------------------------------------------------------------------------------------
  function morePhotonsExist(photons)
    type(photonStream), intent(inout) :: photons
    logical                           :: morePhotonsExist
    morePhotonsExist = photons%currentPhoton > 0 .and. photons%currentPhoton <= photons%numberOfPhotons
  end function morePhotonsExist

  subroutine getNextPhoton(photons, xPosition, yPosition, zPosition, solarMu, solarAzimuth, status)
    type(photonStream), intent(inout) :: photons
    real,               intent(  out) :: xPosition, yPosition, zPosition, solarMu, solarAzimuth
    type(ErrorMessage), intent(inout) :: status
    if(photons%currentPhoton < 1) then
      call setStateToFailure(status, "getNextPhoton: photons have not been initialized.")
    else if(photons%currentPhoton > photons%numberOfPhotons) then
      call setStateToFailure(status, "getNextPhoton: Ran out of photons")
    end if
    if(stateIsFailure(status)) return
    call materialise(photons)
    xPosition    = photons%xPosition (photons%currentPhoton)
    yPosition    = photons%yPosition (photons%currentPhoton)
    zPosition    = photons%zPosition (photons%currentPhoton)
    solarMu      = photons%initialMu (photons%currentPhoton)
    solarAzimuth = photons%initialPhi(photons%currentPhoton)
    photons%currentPhoton = photons%currentPhoton + 1
  end subroutine getNextPhoton

  subroutine finalize_PhotonStream(photons)
    type(photonStream), intent(inout) :: photons
    if(associated(photons%xPosition))  deallocate(photons%xPosition)
    if(associated(photons%yPosition))  deallocate(photons%yPosition)
    if(associated(photons%zPosition))  deallocate(photons%zPosition)
    if(associated(photons%initialMu))  deallocate(photons%initialMu)
    if(associated(photons%initialPhi)) deallocate(photons%initialPhi)
    photons%currentPhoton = 0; photons%numberOfPhotons = 0; photons%lazyDirectional = .false.
  end subroutine finalize_PhotonStream

  ! -- what the GPU integrator needs ----------------------------------------------------------------
  subroutine describeStream(photons, remaining, lazyDirectional, solarMu, solarAzimuthDegrees)
    type(photonStream), intent(in ) :: photons
    integer,            intent(out) :: remaining
    logical,            intent(out) :: lazyDirectional
    real,               intent(out) :: solarMu, solarAzimuthDegrees
    remaining = 0
    if(photons%currentPhoton > 0) remaining = max(photons%numberOfPhotons - photons%currentPhoton + 1, 0)
    lazyDirectional     = photons%lazyDirectional
    solarMu             = photons%solarMu
    solarAzimuthDegrees = photons%solarAzimuthDegrees
  end subroutine describeStream

  ! remaining photons of an explicit stream (pointers into the stream's own storage)
  subroutine streamArrays(photons, x, y, z, mu, phi)
    type(photonStream), intent(in) :: photons
    real, dimension(:), pointer    :: x, y, z, mu, phi
    x   => photons%xPosition (photons%currentPhoton:)
    y   => photons%yPosition (photons%currentPhoton:)
    z   => photons%zPosition (photons%currentPhoton:)
    mu  => photons%initialMu (photons%currentPhoton:)
    phi => photons%initialPhi(photons%currentPhoton:)
  end subroutine streamArrays

  subroutine consumeStream(photons)
    type(photonStream), intent(inout) :: photons
    photons%currentPhoton = photons%numberOfPhotons + 1
  end subroutine consumeStream
end module monteCarloIllumination
