! Fortran-95 shell of the MI355X photon-tracing integrator -- random numbers.
! Public interface of the reference's module RandomNumbers (Code/RandomNumbersForMC.f95:99-110):
! randomNumberSequence, new_RandomNumberSequence(seed) with scalar or vector seed, getRandomInt,
! getRandomPositiveInt, getRandomDouble, getRandomReal, finalize_RandomNumberSequence.
!
! Host side: a Mersenne Twister (MT19937, Matsumoto & Nishimura 1998, init_genrand / init_by_array) so that
! host-generated photon streams see the same deviates as with the reference.  Device side: the seed words
! are kept (getSeedWords) and key the per-photon Philox4x32-10 streams of the HIP kernels, the driver's
! seed = (/iseed, batch/) (Example-Drivers/monteCarloDriver.f95:277) becoming the Philox key.
module RandomNumbers
  implicit none
  private
  integer, parameter :: i8 = selected_int_kind(18)
  integer, parameter :: nWords = 624, lag = 397
  integer(i8), parameter :: mask32 = 4294967295_i8, upperBit = 2147483648_i8, lowerBits = 2147483647_i8
  integer(i8), parameter :: matrixA = 2567483615_i8

  type randomNumberSequence
    private
    integer(i8), dimension(0:nWords - 1) :: mt = 0_i8     ! 32-bit words held in 64-bit integers
    integer                              :: next = nWords
    integer                              :: seedWord0 = 0, seedWord1 = 0
    integer(i8)                          :: photonsDrawn = 0_i8   ! photons whose Philox streams this sequence has handed out
  end type randomNumberSequence

  interface new_RandomNumberSequence
    module procedure seedFromScalar, seedFromVector
  end interface new_RandomNumberSequence

  public :: randomNumberSequence
  public :: new_RandomNumberSequence, finalize_RandomNumberSequence, &
            getRandomInt, getRandomPositiveInt, getRandomReal, getRandomDouble
  public :: getSeedWords, reservePhotonStreams   ! extensions used by the GPU integrator
contains
  function seedFromScalar(seed) result(twister)
    integer, intent(in)        :: seed
    type(randomNumberSequence) :: twister
    call initGenrand(twister, toUnsigned(seed))
    twister%seedWord0 = seed
    twister%seedWord1 = 0
    twister%photonsDrawn = 0_i8
  end function seedFromScalar

  function seedFromVector(seed) result(twister)
    integer, dimension(0:), intent(in) :: seed
    type(randomNumberSequence)         :: twister
    integer     :: i, j, k, n
    integer(i8) :: prev

    n = size(seed)
    call initGenrand(twister, 19650218_i8)
    i = 1; j = 0
    do k = max(nWords, n), 1, -1
      prev = twister%mt(i - 1)
      twister%mt(i) = iand(ieor(twister%mt(i), iand(ieor(prev, ishft(prev, -30)) * 1664525_i8, mask32)) &
                           + toUnsigned(seed(j)) + int(j, i8), mask32)
      i = i + 1; j = j + 1
      if(i >= nWords) then
        twister%mt(0) = twister%mt(nWords - 1); i = 1
      end if
      if(j >= n) j = 0
    end do
    do k = nWords - 1, 1, -1
      prev = twister%mt(i - 1)
      twister%mt(i) = iand(ieor(twister%mt(i), iand(ieor(prev, ishft(prev, -30)) * 1566083941_i8, mask32)) &
                           - int(i, i8) + 4294967296_i8, mask32)
      i = i + 1
      if(i >= nWords) then
        twister%mt(0) = twister%mt(nWords - 1); i = 1
      end if
    end do
    twister%mt(0) = upperBit
    twister%next  = nWords
    twister%seedWord0 = seed(0)
    twister%seedWord1 = 0
    if(n > 1) twister%seedWord1 = seed(1)
    do k = 2, n - 1    ! longer seed vectors are folded into the second key word
      twister%seedWord1 = ieor(twister%seedWord1 * 31, seed(k))
    end do
  end function seedFromVector

  subroutine finalize_RandomNumberSequence(twister)
    type(randomNumberSequence), intent(out) :: twister
    twister%next = nWords
    twister%mt(:) = 0_i8
    twister%photonsDrawn = 0_i8
  end subroutine finalize_RandomNumberSequence

  subroutine getSeedWords(twister, word0, word1)
    type(randomNumberSequence), intent(in ) :: twister
    integer,                    intent(out) :: word0, word1
    word0 = twister%seedWord0
    word1 = twister%seedWord1
  end subroutine getSeedWords

  ! The GPU integrator gives photon i of a sequence the Philox stream (key = seed words, counter = i).  A sequence that
  ! is used for a second computeRadiativeTransfer without being re-seeded must not hand out the same streams again
  ! (the reference's Mersenne Twister simply goes on): every call reserves the next n photon numbers.
  subroutine reservePhotonStreams(twister, n, first)
    type(randomNumberSequence), intent(inout) :: twister
    integer,                    intent(in   ) :: n
    integer(i8),                intent(  out) :: first
    first = twister%photonsDrawn
    twister%photonsDrawn = twister%photonsDrawn + int(n, i8)
  end subroutine reservePhotonStreams

  ! -- generator ------------------------------------------------------------------------------------
  pure function toUnsigned(k) result(u)
    integer, intent(in) :: k
    integer(i8)         :: u
    u = int(k, i8)
    if(u < 0_i8) u = u + 4294967296_i8
  end function toUnsigned

  subroutine initGenrand(twister, s)
    type(randomNumberSequence), intent(inout) :: twister
    integer(i8),                intent(in   ) :: s
    integer     :: i
    integer(i8) :: prev
    twister%mt(0) = iand(s, mask32)
    do i = 1, nWords - 1
      prev = twister%mt(i - 1)
      ! 1812433253 * x needs 63 bits at most: the product of two numbers below 2**32 / 2**31
      twister%mt(i) = iand(iand(1812433253_i8 * ieor(prev, ishft(prev, -30)), mask32) + int(i, i8), mask32)
    end do
    twister%next = nWords
  end subroutine initGenrand

  subroutine reload(twister)
    type(randomNumberSequence), intent(inout) :: twister
    integer     :: k
    integer(i8) :: y
    do k = 0, nWords - 1
      y = ior(iand(twister%mt(k), upperBit), iand(twister%mt(mod(k + 1, nWords)), lowerBits))
      twister%mt(k) = ieor(twister%mt(mod(k + lag, nWords)), ishft(y, -1))
      if(iand(y, 1_i8) /= 0_i8) twister%mt(k) = ieor(twister%mt(k), matrixA)
    end do
    twister%next = 0
  end subroutine reload

  function nextWord(twister) result(y)
    type(randomNumberSequence), intent(inout) :: twister
    integer(i8)                               :: y
    if(twister%next >= nWords) call reload(twister)
    y = twister%mt(twister%next)
    twister%next = twister%next + 1
    y = ieor(y, ishft(y, -11))
    y = ieor(y, iand(ishft(y, 7),  2636928640_i8))
    y = ieor(y, iand(ishft(y, 15), 4022730752_i8))
    y = iand(ieor(y, ishft(y, -18)), mask32)
  end function nextWord

  ! Random integer in [-2**31, 2**31) (the unsigned 32-bit word reinterpreted), as the reference returns it.
  function getRandomInt(twister)
    type(randomNumberSequence), intent(inout) :: twister
    integer                                   :: getRandomInt
    integer(i8) :: y
    y = nextWord(twister)
    if(y >= upperBit) y = y - 4294967296_i8
    getRandomInt = int(y)
  end function getRandomInt

  function getRandomPositiveInt(twister)
    type(randomNumberSequence), intent(inout) :: twister
    integer                                   :: getRandomPositiveInt
    getRandomPositiveInt = int(ishft(nextWord(twister), -1))
  end function getRandomPositiveInt

  ! Uniform on [0, 1] with 32-bit resolution: word / (2**32 - 1)
  function getRandomDouble(twister)
    type(randomNumberSequence), intent(inout) :: twister
    double precision                          :: getRandomDouble
    getRandomDouble = dble(nextWord(twister)) / 4294967295.0d0
  end function getRandomDouble

  function getRandomReal(twister)
    type(randomNumberSequence), intent(inout) :: twister
    real                                      :: getRandomReal
    getRandomReal = real(getRandomDouble(twister))
  end function getRandomReal
end module RandomNumbers
