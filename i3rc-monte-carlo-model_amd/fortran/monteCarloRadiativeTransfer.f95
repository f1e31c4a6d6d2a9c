! Fortran-95 shell of the MI355X photon-tracing integrator -- the integrator object.
! Public interface of the reference's module monteCarloRadiativeTransfer
! (Integrators/monteCarloRadiativeTransfer.f95:154-156): integrator, new_Integrator, copy_Integrator,
! isReady_Integrator, finalize_Integrator, specifyParameters (same 16 optional keywords), computeRadiativeTransfer,
! reportResults -- so Example-Drivers/monteCarloDriver.f95 and planeParallel.f95 compile and link unchanged.
!
! What differs from the reference is WHERE the photon loop runs: computeRT (:400-707) and everything it calls are
! HIP kernels on an MI355X, reached through the C ABI of include/i3rc_hip.h.  This module keeps the host-side
! duties: problem set-up (property grids by component :217-234, inverse / forward / hybrid phase-function tables
! :1809-2039), parameter validation with the reference's messages (:872-947), normalisation (:353-395, done in
! float64 by i3rc_hip_normalise) and reporting (:711-826).
module monteCarloRadiativeTransfer
  use, intrinsic :: iso_c_binding
  use CharacterUtils,           only: IntToChar
  use ErrorMessages,            only: ErrorMessage, stateIsFailure, setStateToFailure, setStateToWarning, &
                                      setStateToSuccess, setStateToCompleteSuccess
  use RandomNumbers,            only: randomNumberSequence, getRandomReal, getSeedWords, reservePhotonStreams
  use numericUtilities,         only: findIndex
  use scatteringPhaseFunctions, only: phaseFunctionTable, getInfo_PhaseFunctionTable, copy_PhaseFunctionTable, &
                                      getPhaseFunctionValues, finalize_PhaseFunctionTable
  use inversePhaseFunctions,    only: computeInversePhaseFuncTable
  use opticalProperties,        only: domain, getInfo_Domain, getOpticalPropertiesByComponent
  use surfaceProperties,        only: surfaceDescription, copy_surfaceDescription, finalize_surfaceDescription, &
                                      isReady_surfaceDescription, getSurfaceGrid
  use monteCarloIllumination,   only: photonStream, morePhotonsExist, describeStream, streamArrays, consumeStream
  use MultipleProcesses,        only: localDevice, sumAcrossProcesses
  use i3rcHipInterface
  implicit none
  private

  integer, parameter :: smallestTableSize = 9001
  real,    parameter :: defaultHybridWidth = 7., largestHybridWidth = 30.
  real,    parameter :: Pi = 3.14159265358979312

  type integrator
    private
    logical     :: readyToCompute = .false., computeIntensity = .false.
    integer     :: minForwardTableSize = smallestTableSize, minInverseTableSize = smallestTableSize
    type(c_ptr) :: device = c_null_ptr                         ! handle of the HIP integrator
    integer     :: deviceIndex = 0
    ! host copies (copy_Integrator rebuilds a device instance from them)
    real,    dimension(:),          pointer :: xPosition => null(), yPosition => null(), zPosition => null()
    real,    dimension(:, :, :),    pointer :: totalExt           => null()
    real,    dimension(:, :, :, :), pointer :: cumulativeExt      => null(), ssa => null()
    integer, dimension(:, :, :, :), pointer :: phaseFunctionIndex => null()
    type(phaseFunctionTable), dimension(:), pointer :: forwardTables => null()
    integer, dimension(:), pointer :: inverseSizeOnDevice => null(), forwardSizeOnDevice => null()
    logical                       :: forwardTablesStale = .true.
    type(i3rc_params)             :: parameters
    real                          :: hybridPhaseFunWidth = defaultHybridWidth
    logical                       :: useSurfaceBDRF = .false.
    type(surfaceDescription)      :: surfaceBDRF
    real, dimension(:, :), pointer :: intensityDirections => null()
    ! results of the last computeRadiativeTransfer
    real, dimension(:, :),       pointer :: fluxUp => null(), fluxDown => null(), fluxAbsorbed => null()
    real, dimension(:, :, :),    pointer :: volumeAbsorption => null(), intensity => null()
    real, dimension(:, :, :, :), pointer :: intensityByComponent => null()
    double precision :: photonsProcessed = 0.d0, photonsDropped = 0.d0
    ! raw tallies of the batches of the last computeRadiativeTransferBatches (one column per batch)
    real(c_double), dimension(:, :), pointer :: batchTallies => null()
    ! ... of which the first batchesFetched have arrived (a streamed loop: see computeRadiativeTransferBatches), and the loop itself
    integer :: batchesFetched = 0, batchSeed0 = 0, batchFirst = 0, batchPhotons = 0
    real    :: batchMu = 0., batchAzimuth = 0.
    logical :: resultsValid = .false.      ! reportResults has a batch to report (not so between announcing a streamed loop and its first selectBatchResults)
    ! sums and sums of squares over the batches of the last computeRadiativeTransferBatchMoments (i3rc_moments_layout)
    real(c_double), dimension(:), pointer :: momentSums => null(), momentSquares => null()
    integer :: momentBatches = 0
  end type integrator

  public :: integrator
  public :: new_Integrator, copy_Integrator, isReady_Integrator, finalize_Integrator, &
            specifyParameters, computeRadiativeTransfer, reportResults
  ! Not in the reference: a driver's batch loop as one call, the batches overlapping on the device (see the procedures)
  public :: computeRadiativeTransferBatches, selectBatchResults
  ! ... and the same loop with its statistics -- all a driver keeps of its batches -- gathered on the device
  public :: computeRadiativeTransferBatchMoments, reportBatchMoments, sumBatchMomentsAcrossProcesses
contains
  ! ------------------------------------------------------------------------------------------------
  ! Creation
  ! ------------------------------------------------------------------------------------------------
  function new_Integrator(atmosphere, status) result(new)
    type(domain),       intent(in   ) :: atmosphere
    type(ErrorMessage), intent(inout) :: status
    type(integrator)                  :: new
    integer :: nx, ny, nz, nc

    call getInfo_Domain(atmosphere, numX = nx, numY = ny, numZ = nz, numberOfComponents = nc, status = status)
    if(.not. stateIsFailure(status)) then
      allocate(new%xPosition(nx + 1), new%yPosition(ny + 1), new%zPosition(nz + 1))
      call getInfo_Domain(atmosphere, xPosition = new%xPosition, yPosition = new%yPosition, zPosition = new%zPosition, &
                          status = status)
    end if
    if(.not. stateIsFailure(status)) then
      allocate(new%totalExt(nx, ny, nz), new%cumulativeExt(nx, ny, nz, nc), new%ssa(nx, ny, nz, nc), &
               new%phaseFunctionIndex(nx, ny, nz, nc), new%forwardTables(nc))
      call getOpticalPropertiesByComponent(atmosphere, new%totalExt, new%cumulativeExt, new%ssa, &
                                           new%phaseFunctionIndex, new%forwardTables, status)
    end if
    if(stateIsFailure(status)) then
      call setStateToFailure(status, "new_Integrator: Problems reading domain.")
      return
    end if
    ! a deviate of exactly 1 must still select the last component
    where(abs(new%cumulativeExt(:, :, :, nc) - 1.) <= spacing(1.)) new%cumulativeExt(:, :, :, nc) = 1. + spacing(1.)
    new%deviceIndex = localDevice()     ! one process per GPU: the device of this rank
    call createDeviceInstance(new, status)
    if(stateIsFailure(status)) return
    allocate(new%fluxUp(nx, ny), new%fluxDown(nx, ny), new%fluxAbsorbed(nx, ny), new%volumeAbsorption(nx, ny, nz))
    new%fluxUp = 0.; new%fluxDown = 0.; new%fluxAbsorbed = 0.; new%volumeAbsorption = 0.
    new%readyToCompute = .true.
    call setStateToSuccess(status)
  end function new_Integrator

  subroutine createDeviceInstance(this, status)
    type(integrator),   intent(inout) :: this
    type(ErrorMessage), intent(inout) :: status
    integer :: nc
    nc = size(this%cumulativeExt, 4)
    if(i3rc_hip_create(this%device, this%deviceIndex, size(this%totalExt, 1), size(this%totalExt, 2), size(this%totalExt, 3), &
                       nc, this%xPosition, this%yPosition, this%zPosition, this%totalExt, this%cumulativeExt, this%ssa, &
                       this%phaseFunctionIndex) /= 0) then
      call setStateToFailure(status, "new_Integrator: " // trim(lastErrorText(c_null_ptr)))
      this%device = c_null_ptr
      return
    end if
    if(associated(this%inverseSizeOnDevice)) deallocate(this%inverseSizeOnDevice, this%forwardSizeOnDevice)
    allocate(this%inverseSizeOnDevice(nc), this%forwardSizeOnDevice(nc))
    this%inverseSizeOnDevice = 0; this%forwardSizeOnDevice = 0
    this%forwardTablesStale = .true.
  end subroutine createDeviceInstance

  logical function deviceCall(this, rc, where, status)
    type(integrator),   intent(in   ) :: this
    integer,            intent(in   ) :: rc
    character(len = *), intent(in   ) :: where
    type(ErrorMessage), intent(inout) :: status
    deviceCall = rc == 0
    if(.not. deviceCall) call setStateToFailure(status, where // ": " // trim(lastErrorText(this%device)))
  end function deviceCall

  function isReady_Integrator(thisIntegrator)
    type(integrator), intent(in) :: thisIntegrator
    logical                      :: isReady_Integrator
    isReady_Integrator = thisIntegrator%readyToCompute .and. c_associated(thisIntegrator%device)
  end function isReady_Integrator

  ! ------------------------------------------------------------------------------------------------
  ! Parameters
  ! ------------------------------------------------------------------------------------------------
  subroutine specifyParameters(thisIntegrator, surfaceAlbedo, surfaceBDRF,             &
                               minForwardTableSize, minInverseTableSize,               &
                               intensityMus, intensityPhis, computeIntensity,          &
                               useRayTracing,  useRussianRoulette,                     &
                               useRussianRouletteForIntensity, zetaMin,                &
                               useHybridPhaseFunsForIntenCalcs, hybridPhaseFunWidth,   &
                               numOrdersOrigPhaseFunIntenCalcs,                        &
                               limitIntensityContributions, maxIntensityContribution,  &
                               status)
    type(integrator),                   intent(inout) :: thisIntegrator
    real,                     optional, intent(in   ) :: surfaceAlbedo
    type(surfaceDescription), optional, intent(in   ) :: surfaceBDRF
    integer,                  optional, intent(in   ) :: minForwardTableSize, minInverseTableSize
    real, dimension(:),       optional, intent(in   ) :: intensityMus, intensityPhis
    logical,                  optional, intent(in   ) :: computeIntensity, useRayTracing, useRussianRoulette
    logical,                  optional, intent(in   ) :: useRussianRouletteForIntensity
    real,                     optional, intent(in   ) :: zetaMin
    logical,                  optional, intent(in   ) :: useHybridPhaseFunsForIntenCalcs
    real,                     optional, intent(in   ) :: hybridPhaseFunWidth
    integer,                  optional, intent(in   ) :: numOrdersOrigPhaseFunIntenCalcs
    logical,                  optional, intent(in   ) :: limitIntensityContributions
    real,                     optional, intent(in   ) :: maxIntensityContribution
    type(ErrorMessage),                 intent(inout) :: status
    integer :: i, nDir, nx, ny, nc
    real, dimension(:),    pointer :: xs, ys
    real, dimension(:, :), pointer :: albedoGrid
    real, dimension(:), allocatable :: flatDirections

    ! ---- checks (messages as in the reference :872-947)
    if(present(surfaceBDRF) .and. present(surfaceAlbedo)) &
      call setStateToFailure(status, "specifyParameters: only one surface specification can be provided")
    if(present(surfaceAlbedo)) then
      if(surfaceAlbedo > 1. .or. surfaceAlbedo < 0.) &
        call setStateToFailure(status, "specifyParameters: surface albedo out of range.")
    end if
    if(present(surfaceBDRF)) then
      if(.not. isReady_surfaceDescription(surfaceBDRF)) &
        call setStateToFailure(status, "specifyParameters: surface description isn't valid.")
    end if
    if(present(minForwardTableSize)) then
      if(minForwardTableSize < smallestTableSize) &
        call setStateToWarning(status, "specifyParameters: minForwardTableSize less than default. Value ignored.")
    end if
    if(present(minInverseTableSize)) then
      if(minInverseTableSize < smallestTableSize) &
        call setStateToWarning(status, "specifyParameters: minInverseTableSize less than default. Value ignored.")
    end if
    if(present(hybridPhaseFunWidth)) then
      if(hybridPhaseFunWidth > largestHybridWidth .or. hybridPhaseFunWidth < 0.) &
        call setStateToWarning(status, "specifyParameters: hybridPhaseFunWidth out of range (0 to " //         &
                               trim(IntToChar(int(largestHybridWidth))) // "degrees)." // "Using default (" // &
                               trim(IntToChar(int(defaultHybridWidth))) // ")")
    end if
    if(present(numOrdersOrigPhaseFunIntenCalcs)) then
      if(numOrdersOrigPhaseFunIntenCalcs < 0) &
        call setStateToWarning(status, "specifyParameters: numOrdersExactPhaseFunIntenCalcs less than 0." // "Using default (0)")
    end if
    if(present(maxIntensityContribution)) then
      if(maxIntensityContribution <= 0.) &
        call setStateToWarning(status, "specifyParameters: maxIntensityContribution <= 0. Value is unchanged.")
    end if
    if(present(intensityMus) .neqv. present(intensityPhis)) &
      call setStateToFailure(status, "specifyParameters: Both or neither of intensityMus and intensityPhis must be supplied")
    if(present(intensityMus) .and. present(intensityPhis)) then
      if(size(intensityMus) /= size(intensityPhis)) then
        call setStateToFailure(status, "specifyParameters: intensityMus, intensityPhis must be the same length.")
      else
        if(any(intensityMus < -1.) .or. any(intensityMus > 1.)) &
          call setStateToFailure(status, "specifyParameters: intensityMus must be between -1 and 1")
        if(any(abs(intensityMus) < tiny(intensityMus))) &
          call setStateToFailure(status, "specifyParameters: intensityMus can't be 0 (directly sideways)")
        if(any(intensityPhis < 0.) .or. any(intensityPhis > 360.)) &
          call setStateToFailure(status, "specifyParameters: intensityPhis must be between 0 and 360")
        if(size(intensityMus) > I3RC_MAX_DIRECTIONS) &
          call setStateToFailure(status, "specifyParameters: at most 20 intensity directions")
      end if
    end if
    if(present(computeIntensity)) then
      if(.not. computeIntensity .and. present(intensityMus)) &
        call setStateToWarning(status, "specifyParameters: intensity directions *and* computeIntensity set to false." // &
                                       "Will compute intensity at given angles.")
      if(computeIntensity .and. .not. present(intensityMus) .and. .not. associated(thisIntegrator%intensityDirections)) &
        call setStateToFailure(status, "specifyParameters: Can't compute intensity without specifying directions.")
    end if
    if(.not. c_associated(thisIntegrator%device)) &
      call setStateToFailure(status, "specifyParameters: integrator hasn't been initialized.")
    if(stateIsFailure(status)) return

    ! ---- store
    nx = size(thisIntegrator%totalExt, 1); ny = size(thisIntegrator%totalExt, 2); nc = size(thisIntegrator%cumulativeExt, 4)
    if(present(surfaceAlbedo)) then
      thisIntegrator%parameters%surfaceAlbedo  = surfaceAlbedo
      thisIntegrator%parameters%useSurfaceBDRF = 0
      thisIntegrator%useSurfaceBDRF = .false.
    else if(present(surfaceBDRF)) then
      call finalize_surfaceDescription(thisIntegrator%surfaceBDRF)
      thisIntegrator%surfaceBDRF = copy_surfaceDescription(surfaceBDRF)
      call getSurfaceGrid(thisIntegrator%surfaceBDRF, xs, ys, albedoGrid)
      if(.not. deviceCall(thisIntegrator, i3rc_hip_set_surface(thisIntegrator%device, size(xs) - 1, size(ys) - 1, xs, ys, &
                          reshape(albedoGrid, (/ size(albedoGrid) /))), "specifyParameters", status)) return
      thisIntegrator%parameters%useSurfaceBDRF = 1
      thisIntegrator%useSurfaceBDRF = .true.
    end if
    if(present(useRayTracing))       thisIntegrator%parameters%useRayTracing      = merge(1, 0, useRayTracing)
    if(present(minForwardTableSize)) thisIntegrator%minForwardTableSize = max(minForwardTableSize, smallestTableSize)
    if(present(minInverseTableSize)) thisIntegrator%minInverseTableSize = max(minInverseTableSize, smallestTableSize)
    if(present(useRussianRoulette))  thisIntegrator%parameters%useRussianRoulette = merge(1, 0, useRussianRoulette)
    if(present(useRussianRouletteForIntensity)) &
      thisIntegrator%parameters%useRussianRouletteForIntensity = merge(1, 0, useRussianRouletteForIntensity)
    if(present(zetaMin)) then
      if(zetaMin < 0.) then
        call setStateToWarning(status, "specifyParameters: zetaMin must be >= 0. Value is unchanged.")
      else
        thisIntegrator%parameters%zetaMin = zetaMin
        if(zetaMin > 1.) call setStateToWarning(status, "specifyParameters: zetaMin > 1. That's kind of large.")
      end if
    end if
    if(present(useHybridPhaseFunsForIntenCalcs)) then
      thisIntegrator%parameters%useHybridPhaseFunsForIntenCalcs = merge(1, 0, useHybridPhaseFunsForIntenCalcs)
      thisIntegrator%forwardTablesStale = .true.
    end if
    if(present(hybridPhaseFunWidth)) then
      if(hybridPhaseFunWidth > 0. .and. hybridPhaseFunWidth < largestHybridWidth) then
        thisIntegrator%hybridPhaseFunWidth = hybridPhaseFunWidth
      else
        thisIntegrator%hybridPhaseFunWidth = defaultHybridWidth
      end if
      thisIntegrator%forwardTablesStale = .true.     ! the hybrid tables must be rebuilt
    end if
    if(present(numOrdersOrigPhaseFunIntenCalcs)) &
      thisIntegrator%parameters%numOrdersOrigPhaseFunIntenCalcs = max(numOrdersOrigPhaseFunIntenCalcs, 0)
    if(present(limitIntensityContributions)) &
      thisIntegrator%parameters%limitIntensityContributions = merge(1, 0, limitIntensityContributions)
    if(present(maxIntensityContribution)) then
      if(maxIntensityContribution > 0.) thisIntegrator%parameters%maxIntensityContribution = maxIntensityContribution
    end if

    if(present(intensityMus) .and. present(intensityPhis)) then
      nDir = size(intensityMus)
      if(associated(thisIntegrator%intensityDirections))  deallocate(thisIntegrator%intensityDirections)
      if(associated(thisIntegrator%intensity))            deallocate(thisIntegrator%intensity)
      if(associated(thisIntegrator%intensityByComponent)) deallocate(thisIntegrator%intensityByComponent)
      allocate(thisIntegrator%intensityDirections(3, nDir), thisIntegrator%intensity(nx, ny, nDir), &
               thisIntegrator%intensityByComponent(nx, ny, nDir, 0:nc), flatDirections(3 * nDir))
      do i = 1, nDir
        thisIntegrator%intensityDirections(:, i) = directionCosines(intensityMus(i), intensityPhis(i) * Pi / 180.)
      end do
      thisIntegrator%intensity = 0.; thisIntegrator%intensityByComponent = 0.
      flatDirections = reshape(thisIntegrator%intensityDirections, (/ 3 * nDir /))
      if(.not. deviceCall(thisIntegrator, i3rc_hip_set_directions(thisIntegrator%device, nDir, flatDirections), &
                          "specifyParameters", status)) return
      deallocate(flatDirections)
      thisIntegrator%computeIntensity = .true.
    end if
    if(present(computeIntensity)) then
      if(.not. computeIntensity .and. .not. present(intensityMus)) then
        if(associated(thisIntegrator%intensityDirections))  deallocate(thisIntegrator%intensityDirections)
        if(associated(thisIntegrator%intensity))            deallocate(thisIntegrator%intensity)
        if(associated(thisIntegrator%intensityByComponent)) deallocate(thisIntegrator%intensityByComponent)
        allocate(flatDirections(3))
        flatDirections = 0.
        if(.not. deviceCall(thisIntegrator, i3rc_hip_set_directions(thisIntegrator%device, 0, flatDirections), &
                            "specifyParameters", status)) return
        deallocate(flatDirections)
        thisIntegrator%computeIntensity = .false.
      end if
    end if
    if(.not. deviceCall(thisIntegrator, i3rc_hip_set_params(thisIntegrator%device, thisIntegrator%parameters), &
                        "specifyParameters", status)) return
    call setStateToSuccess(status)
  end subroutine specifyParameters

  pure function directionCosines(mu, phi) result(s)
    real, intent(in)   :: mu, phi
    real, dimension(3) :: s
    real :: sinTheta
    sinTheta = sqrt(1. - mu**2)
    s = (/ sinTheta * cos(phi), sinTheta * sin(phi), mu /)
  end function directionCosines

  ! ------------------------------------------------------------------------------------------------
  ! Phase-function tables for the device
  ! ------------------------------------------------------------------------------------------------
  subroutine ensureTables(this, status)
    type(integrator),   intent(inout) :: this
    type(ErrorMessage), intent(inout) :: status
    integer :: c, nEntries, nSteps, j
    real, dimension(:, :), allocatable :: table, hybrid
    real, dimension(:),    allocatable :: angles

    do c = 1, size(this%forwardTables)
      call getInfo_PhaseFunctionTable(this%forwardTables(c), nEntries = nEntries, status = status)
      if(stateIsFailure(status)) exit
      if(this%inverseSizeOnDevice(c) < this%minInverseTableSize) then
        nSteps = this%minInverseTableSize
        allocate(table(nSteps, nEntries))
        call computeInversePhaseFuncTable(this%forwardTables(c), table, status)
        if(.not. stateIsFailure(status)) then
          if(deviceCall(this, i3rc_hip_set_inverse_table(this%device, c, nSteps, nEntries, reshape(table, (/ size(table) /))), &
                        "tabulateInversePhaseFunctions", status)) this%inverseSizeOnDevice(c) = nSteps
        end if
        deallocate(table)
        if(stateIsFailure(status)) exit
      end if
      if(this%computeIntensity .and. (this%forwardSizeOnDevice(c) < this%minForwardTableSize .or. this%forwardTablesStale)) then
        nSteps = this%minForwardTableSize
        allocate(table(nSteps, nEntries), hybrid(nSteps, nEntries), angles(nSteps))
        angles(:) = (/ (j, j = 0, nSteps - 1) /) / real(nSteps - 1) * Pi
        call getPhaseFunctionValues(this%forwardTables(c), angles, table, status)
        hybrid = table
        if(this%parameters%useHybridPhaseFunsForIntenCalcs /= 0 .and. this%hybridPhaseFunWidth > 0.) &
          call spliceGaussianPeak(angles, table, this%hybridPhaseFunWidth, hybrid)
        if(.not. stateIsFailure(status)) then
          if(deviceCall(this, i3rc_hip_set_forward_tables(this%device, c, nSteps, nEntries, reshape(hybrid, (/ size(hybrid) /)), &
                        reshape(table, (/ size(table) /))), "tabulateForwardPhaseFunctions", status)) &
            this%forwardSizeOnDevice(c) = nSteps
        end if
        deallocate(table, hybrid, angles)
        if(stateIsFailure(status)) exit
      end if
    end do
    if(stateIsFailure(status)) then
      call setStateToFailure(status, "tabulatePhaseFunctions: failed on component" // trim(IntToChar(c)))
    else
      this%forwardTablesStale = .false.
      call setStateToSuccess(status)
    end if
  end subroutine ensureTables

  ! Hybrid phase functions for the local estimate: a Gaussian of the given width replaces the forward peak,
  ! joined where the normalised Gaussian meets the original (first sign change found by doubling steps from the
  ! width, then bisection); entries without a crossing keep the original.
  subroutine spliceGaussianPeak(angles, original, widthDegrees, spliced)
    real, dimension(:),    intent(in ) :: angles
    real, dimension(:, :), intent(in ) :: original
    real,                  intent(in ) :: widthDegrees
    real, dimension(:, :), intent(out) :: spliced
    real, dimension(size(angles)) :: cosines, gaussian
    integer :: n, e, lo, hi, mid, stride
    real    :: dLo, dHi, dMid, scale
    logical :: bracketed

    n = size(angles)
    cosines  = cos(angles)
    gaussian = exp(-(angles / (widthDegrees * Pi / 180))**2)
    spliced  = original
    do e = 1, size(original, 2)
      lo = findIndex(widthDegrees * Pi / 180., angles) + 1
      if(lo >= n - 2) exit
      dLo = mismatch(lo)
      stride = 1
      bracketed = .false.
      do
        hi = min(lo + stride, n - 1)
        dHi = mismatch(hi)
        if(lo == n - 1) exit
        if(dLo * dHi < 0) then
          bracketed = .true.
          exit
        end if
        lo = hi; dLo = dHi; stride = 2 * stride
      end do
      if(.not. bracketed) cycle
      do while(hi > lo + 1)
        mid = (lo + hi) / 2
        dMid = mismatch(mid)
        if(dMid * dHi < 0) then
          lo = mid; dLo = dMid
        else
          hi = mid; dHi = dMid
        end if
      end do
      scale = peakScale(lo)
      spliced(:lo, e)     = scale * gaussian(:lo)
      spliced(lo + 1:, e) = original(lo + 1:, e)
    end do
  contains
    ! factor that keeps the spliced function normalised (integral over mu = 2) when joined at index k
    real function peakScale(k)
      integer, intent(in) :: k
      real :: areaGaussian, areaOriginal
      areaGaussian = dot_product(0.5 * (gaussian(1:k - 1) + gaussian(2:k)), cosines(1:k - 1) - cosines(2:k))
      areaOriginal = dot_product(0.5 * (original(k:n - 1, e) + original(k + 1:n, e)), cosines(k:n - 1) - cosines(k + 1:n))
      if(areaOriginal >= 2.0) then
        peakScale = 1.0 / areaGaussian
      else
        peakScale = (2. - areaOriginal) / areaGaussian
      end if
    end function peakScale
    real function mismatch(k)
      integer, intent(in) :: k
      mismatch = peakScale(k) * gaussian(k) - original(k, e)
    end function mismatch
  end subroutine spliceGaussianPeak

  ! ------------------------------------------------------------------------------------------------
  ! The computation
  ! ------------------------------------------------------------------------------------------------
  subroutine computeRadiativeTransfer(thisIntegrator, randomNumbers, incomingPhotons, status)
    type(integrator),           intent(inout) :: thisIntegrator
    type(randomNumberSequence), intent(inout) :: randomNumbers
    type(photonStream),         intent(inout) :: incomingPhotons
    type(ErrorMessage),         intent(inout) :: status
    integer :: remaining, seed0, seed1
    integer(selected_int_kind(18)) :: firstPhoton
    logical :: lazy
    real    :: mu0, azimuth, advance
    type(i3rc_source)       :: source
    type(i3rc_tally_layout) :: layout
    real, dimension(:), pointer :: x, y, z, mus, phis
    real(c_float), dimension(:), allocatable, target :: sx, sy, sz, smu, sphi
    real(c_double), dimension(:), allocatable :: raw

    if(.not. isReady_Integrator(thisIntegrator)) then
      call setStateToFailure(status, "computeRadiativeTransfer: problem not completely specified.")
      return
    end if
    call ensureTables(thisIntegrator, status)
    if(stateIsFailure(status)) return

    call describeStream(incomingPhotons, remaining, lazy, mu0, azimuth)
    if(.not. morePhotonsExist(incomingPhotons) .or. remaining < 1) then
      call setStateToFailure(status, "computeRadiativeTransfer: Didn't process any photons.")
      return
    end if
    if(lazy) then
      source%kind = 0; source%solarMu = mu0; source%solarAzimuth = azimuth
    else
      call streamArrays(incomingPhotons, x, y, z, mus, phis)
      allocate(sx(remaining), sy(remaining), sz(remaining), smu(remaining), sphi(remaining))
      sx = x(:remaining); sy = y(:remaining); sz = z(:remaining); smu = mus(:remaining); sphi = phis(:remaining)
      source%kind = 1
      source%x = c_loc(sx); source%y = c_loc(sy); source%z = c_loc(sz); source%mu = c_loc(smu); source%phi = c_loc(sphi)
    end if
    call getSeedWords(randomNumbers, seed0, seed1)
    call reservePhotonStreams(randomNumbers, remaining, firstPhoton)   ! (a sequence used again goes on with fresh photon streams)

    if(.not. deviceCall(thisIntegrator, i3rc_hip_get_tally_layout(thisIntegrator%device, layout), &
                        "computeRadiativeTransfer", status)) return
    allocate(raw(layout%total))
    if(lazy .and. firstPhoton == 0) then
      ! A Directional batch of a fresh sequence -- what a driver's batch loop asks for, once per batch, with
      ! seed = (/iseed, batch/) (monteCarloDriver.f95:277-297).  A call cannot return before the batch's last photon has,
      ! and the last photons of a batch are the few with a thousand scatterings: the library looks ahead -- once two
      ! calls in a row show the loop it traces the following batches behind this one (i3rc_hip_compute_batch), so that
      ! the next call finds its batch under way or done.  I3RC_LOOK_AHEAD = 0 ... 7 in the environment (default 3).
      if(.not. deviceCall(thisIntegrator, i3rc_hip_compute_batch(thisIntegrator%device, int(seed0, c_int32_t),            &
                          int(seed1, c_int32_t), int(remaining, c_int64_t), source, int(lookAheadDepth(remaining), c_int), &
                          raw), "computeRadiativeTransfer", status)) return
    else
      ! tallies are overwritten, not accumulated, on every call (reference :296-309)
      if(.not. deviceCall(thisIntegrator, i3rc_hip_zero_tallies(thisIntegrator%device), "computeRadiativeTransfer", status)) return
      if(.not. deviceCall(thisIntegrator, i3rc_hip_launch_batch(thisIntegrator%device, int(seed0, c_int32_t), int(seed1, c_int32_t), &
                          int(firstPhoton, c_int64_t), int(remaining, c_int64_t), source), "computeRadiativeTransfer", status)) return
      if(.not. deviceCall(thisIntegrator, i3rc_hip_fetch_tallies(thisIntegrator%device, raw), "computeRadiativeTransfer", status)) return
    end if
    if(.not. unpackTallies(thisIntegrator, raw, layout, "computeRadiativeTransfer", status)) return
    deallocate(raw)
    if(allocated(sx)) deallocate(sx, sy, sz, smu, sphi)

    ! side effects the callers rely on: the stream is consumed and the random sequence has advanced
    call consumeStream(incomingPhotons)
    advance = getRandomReal(randomNumbers)
    if(thisIntegrator%photonsProcessed > 0.d0) then
      call setStateToCompleteSuccess(status, "computeRadiativeTransfer: finished with photons")
    else
      call setStateToFailure(status, "computeRadiativeTransfer: Didn't process any photons.")
    end if
  end subroutine computeRadiativeTransfer

  ! batches to trace ahead of a driver's loop: none for batches so long that their tail does not matter
  integer function lookAheadDepth(numberOfPhotons)
    integer, intent(in) :: numberOfPhotons
    integer, save :: depth = -1
    character(len = 16) :: text
    integer :: rc
    if(depth < 0) then
      depth = 3
      call get_environment_variable("I3RC_LOOK_AHEAD", text, status = rc)
      if(rc == 0 .and. len_trim(text) > 0) read(text, *, iostat = rc) depth
      depth = max(0, min(7, depth))
    end if
    lookAheadDepth = depth
    if(numberOfPhotons > 20000000) lookAheadDepth = 0
  end function lookAheadDepth

  ! raw float64 tallies of one batch -> the normalised results reportResults hands out (:353-395)
  function unpackTallies(thisIntegrator, raw, layout, caller, status) result(ok)
    type(integrator),             intent(inout) :: thisIntegrator
    real(c_double), dimension(:), intent(in   ) :: raw
    type(i3rc_tally_layout),      intent(in   ) :: layout
    character(len = *),           intent(in   ) :: caller
    type(ErrorMessage),           intent(inout) :: status
    logical :: ok
    type(c_ptr) :: pIntensity, pByComponent
    pIntensity = c_null_ptr; pByComponent = c_null_ptr
    if(thisIntegrator%computeIntensity) then
      pIntensity   = c_loc(thisIntegrator%intensity(1, 1, 1))
      pByComponent = c_loc(thisIntegrator%intensityByComponent(1, 1, 1, 0))
    end if
    ok = deviceCall(thisIntegrator, i3rc_hip_normalise(thisIntegrator%device, raw, c_loc(thisIntegrator%fluxUp(1, 1)),   &
                    c_loc(thisIntegrator%fluxDown(1, 1)), c_loc(thisIntegrator%fluxAbsorbed(1, 1)),                        &
                    c_loc(thisIntegrator%volumeAbsorption(1, 1, 1)), pIntensity, pByComponent), caller, status)
    if(.not. ok) return
    thisIntegrator%photonsProcessed = raw(layout%counters + 1 + I3RC_CNT_PHOTONS)
    thisIntegrator%photonsDropped   = raw(layout%counters + 1 + I3RC_CNT_DROPPED)
    thisIntegrator%resultsValid     = .true.
  end function unpackTallies

  ! ------------------------------------------------------------------------------------------------
  ! Not in the reference: the batch loop of a driver (Example-Drivers/monteCarloDriver.f95:283-326) as one call.
  !   Batch b = firstBatch ... firstBatch + numBatches - 1 is traced exactly as
  !     computeRadiativeTransfer(integrator, new_RandomNumberSequence(seed = (/ iseed, b /)),
  !                              new_PhotonStream(solarMu, solarAzimuth, numberOfPhotons, ...), status)
  !   traces it (same photons), but several batches share the device at a time (i3rc_hip_run_batches): a batch of 1e5 ... 1e6
  !   photons ends with a long tail -- a few photons with a thousand scatterings -- that the next batch's photons cover.
  !   selectBatchResults(integrator, k, status), k = 1 ... numBatches, then makes batch firstBatch + k - 1 the one
  !   reportResults reports (call it before reportResults: a streamed loop has no results before).  The raw tallies of the
  !   batches taken over so far stay with the integrator until the next call; asking for batch k takes over all up to k.
  ! ------------------------------------------------------------------------------------------------
  subroutine computeRadiativeTransferBatches(thisIntegrator, iseed, firstBatch, numBatches, solarMu, solarAzimuth, &
                                             numberOfPhotons, status, batchesInFlight)
    type(integrator),   intent(inout) :: thisIntegrator
    integer,            intent(in   ) :: iseed, firstBatch, numBatches, numberOfPhotons
    real,               intent(in   ) :: solarMu, solarAzimuth
    type(ErrorMessage), intent(inout) :: status
    integer, optional,  intent(in   ) :: batchesInFlight
    type(i3rc_source)       :: source
    type(i3rc_tally_layout) :: layout
    integer :: inFlight
    integer(c_int) :: accepted

    if(.not. isReady_Integrator(thisIntegrator)) then
      call setStateToFailure(status, "computeRadiativeTransfer: problem not completely specified.")
      return
    end if
    if(numBatches < 1 .or. numberOfPhotons < 1) then
      call setStateToFailure(status, "computeRadiativeTransfer: Didn't process any photons.")
      return
    end if
    call ensureTables(thisIntegrator, status)
    if(stateIsFailure(status)) return
    inFlight = 0
    if(present(batchesInFlight)) inFlight = batchesInFlight
    source%kind = 0; source%solarMu = solarMu; source%solarAzimuth = solarAzimuth
    if(.not. deviceCall(thisIntegrator, i3rc_hip_get_tally_layout(thisIntegrator%device, layout), &
                        "computeRadiativeTransfer", status)) return
    if(associated(thisIntegrator%batchTallies)) deallocate(thisIntegrator%batchTallies)
    allocate(thisIntegrator%batchTallies(layout%total, numBatches))
    thisIntegrator%batchesFetched = 0
    thisIntegrator%batchSeed0 = iseed;   thisIntegrator%batchFirst = firstBatch; thisIntegrator%batchPhotons = numberOfPhotons
    thisIntegrator%batchMu    = solarMu; thisIntegrator%batchAzimuth = solarAzimuth
    ! Flux problems of the common class are STREAMED: the library is told the loop (i3rc_hip_expect_batches) and starts tracing
    ! it at once, group of batches by group (each group one kernel launch); selectBatchResults(k) takes the batches over as
    ! they are asked for, so that the caller works on batch k while the following ones are traced -- and this call returns
    ! without waiting.  Other problems (radiances, BRDF grids, several components) are traced here and now, several batches
    ! on the device at a time (i3rc_hip_run_batches).
    accepted = 0
    if(inFlight /= 1) then
      if(.not. deviceCall(thisIntegrator, i3rc_hip_expect_batches(thisIntegrator%device, int(iseed, c_int32_t),           &
                          int(firstBatch, c_int32_t), int(numBatches, c_int), int(numberOfPhotons, c_int64_t), source,   &
                          accepted), "computeRadiativeTransfer", status)) then
        deallocate(thisIntegrator%batchTallies)
        return
      end if
    end if
    if(accepted == 0) then
      if(.not. deviceCall(thisIntegrator, i3rc_hip_run_batches(thisIntegrator%device, int(iseed, c_int32_t),              &
                          int(firstBatch, c_int32_t), int(numBatches, c_int), int(numberOfPhotons, c_int64_t), source,   &
                          int(inFlight, c_int), thisIntegrator%batchTallies), "computeRadiativeTransfer", status)) then
        deallocate(thisIntegrator%batchTallies)
        return
      end if
      thisIntegrator%batchesFetched = numBatches
      call selectBatchResults(thisIntegrator, numBatches, status)   ! (as after a loop of computeRadiativeTransfer calls: the last batch)
    else
      ! (streamed: nothing has been traced to the end yet.  Whatever an earlier call left in the integrator is not this loop's:
      ! reportResults fails until selectBatchResults has made one of the loop's batches current)
      thisIntegrator%resultsValid = .false.
      call setStateToSuccess(status)
    end if
  end subroutine computeRadiativeTransferBatches

  subroutine selectBatchResults(thisIntegrator, batch, status)
    type(integrator),   intent(inout) :: thisIntegrator
    integer,            intent(in   ) :: batch
    type(ErrorMessage), intent(inout) :: status
    type(i3rc_tally_layout) :: layout
    type(i3rc_source)       :: source
    if(.not. associated(thisIntegrator%batchTallies)) then
      call setStateToFailure(status, "selectBatchResults: no batches have been computed.")
      return
    end if
    if(batch < 1 .or. batch > size(thisIntegrator%batchTallies, 2)) then
      call setStateToFailure(status, "selectBatchResults: no such batch.")
      return
    end if
    if(.not. deviceCall(thisIntegrator, i3rc_hip_get_tally_layout(thisIntegrator%device, layout), "selectBatchResults", status)) return
    ! the loop was announced with the tally layout of its time: a specifyParameters in between (other radiance directions, say)
    ! changes what a batch's block holds and how long it is -- neither the blocks already here nor the rest of the loop fit any more
    if(layout%total /= size(thisIntegrator%batchTallies, 1)) then
      call setStateToFailure(status, "selectBatchResults: the problem has changed since the batches were announced " // &
                                     "(another tally layout); call computeRadiativeTransferBatches again.")
      return
    end if
    ! a streamed loop: take over the batches up to this one (in the loop's order: the library traces them ahead of us)
    source%kind = 0; source%solarMu = thisIntegrator%batchMu; source%solarAzimuth = thisIntegrator%batchAzimuth
    do while(thisIntegrator%batchesFetched < batch)
      if(.not. deviceCall(thisIntegrator, i3rc_hip_compute_batch(thisIntegrator%device, int(thisIntegrator%batchSeed0, c_int32_t),       &
                          int(thisIntegrator%batchFirst + thisIntegrator%batchesFetched, c_int32_t),                                   &
                          int(thisIntegrator%batchPhotons, c_int64_t), source, 3_c_int,                                                 &
                          thisIntegrator%batchTallies(:, thisIntegrator%batchesFetched + 1)), "selectBatchResults", status)) return
      thisIntegrator%batchesFetched = thisIntegrator%batchesFetched + 1
    end do
    if(.not. unpackTallies(thisIntegrator, thisIntegrator%batchTallies(:, batch), layout, "selectBatchResults", status)) return
    if(thisIntegrator%photonsProcessed > 0.d0) then
      call setStateToCompleteSuccess(status, "computeRadiativeTransfer: finished with photons")
    else
      call setStateToFailure(status, "computeRadiativeTransfer: Didn't process any photons.")
    end if
  end subroutine selectBatchResults

  ! ------------------------------------------------------------------------------------------------
  ! Not in the reference: the batch loop of a driver WITH its statistics.  The reference's drivers keep of every batch only the
  !   first two moments of what reportResults returns (Example-Drivers/monteCarloDriver.f95:300-321) and turn them into mean
  !   and standard error (:358-378).  computeRadiativeTransferBatchMoments traces batches firstBatch ... firstBatch + numBatches - 1
  !   exactly as computeRadiativeTransferBatches does, but every batch is normalised (:353-395) and its values and their squares
  !   are added up ON THE DEVICE (i3rc_hip_run_batches_moments): the per-batch tallies -- megabytes per batch on a cloud field --
  !   never come to the host.  reportBatchMoments hands the sums out in the shapes of the drivers' ...Stats arrays
  !   (last dimension: 1 = sum of x, 2 = sum of x**2 over the batches).  reportResults is not served by this call.
  ! ------------------------------------------------------------------------------------------------
  subroutine computeRadiativeTransferBatchMoments(thisIntegrator, iseed, firstBatch, numBatches, solarMu, solarAzimuth, &
                                                  numberOfPhotons, status)
    type(integrator),   intent(inout) :: thisIntegrator
    integer,            intent(in   ) :: iseed, firstBatch, numBatches, numberOfPhotons
    real,               intent(in   ) :: solarMu, solarAzimuth
    type(ErrorMessage), intent(inout) :: status
    type(i3rc_source)         :: source
    type(i3rc_moments_layout) :: layout
    real(c_double)            :: counters(I3RC_NUM_COUNTERS)

    if(.not. isReady_Integrator(thisIntegrator)) then
      call setStateToFailure(status, "computeRadiativeTransfer: problem not completely specified.")
      return
    end if
    if(numBatches < 1 .or. numberOfPhotons < 1) then
      call setStateToFailure(status, "computeRadiativeTransfer: Didn't process any photons.")
      return
    end if
    call ensureTables(thisIntegrator, status)
    if(stateIsFailure(status)) return
    source%kind = 0; source%solarMu = solarMu; source%solarAzimuth = solarAzimuth
    if(.not. deviceCall(thisIntegrator, i3rc_hip_get_moments_layout(thisIntegrator%device, layout), &
                        "computeRadiativeTransfer", status)) return
    if(associated(thisIntegrator%momentSums)) deallocate(thisIntegrator%momentSums, thisIntegrator%momentSquares)
    allocate(thisIntegrator%momentSums(layout%total), thisIntegrator%momentSquares(layout%total))
    thisIntegrator%momentBatches = 0
    if(.not. deviceCall(thisIntegrator, i3rc_hip_run_batches_moments(thisIntegrator%device, int(iseed, c_int32_t),          &
                        int(firstBatch, c_int32_t), int(numBatches, c_int), int(numberOfPhotons, c_int64_t), source,       &
                        thisIntegrator%momentSums, thisIntegrator%momentSquares, counters), "computeRadiativeTransfer", status)) then
      deallocate(thisIntegrator%momentSums, thisIntegrator%momentSquares)
      return
    end if
    thisIntegrator%momentBatches    = numBatches
    thisIntegrator%resultsValid     = .false.      ! (no single batch is current after this call: reportResults says so)
    thisIntegrator%photonsProcessed = counters(1 + I3RC_CNT_PHOTONS)
    thisIntegrator%photonsDropped   = counters(1 + I3RC_CNT_DROPPED)
    if(thisIntegrator%photonsProcessed > 0.d0) then
      call setStateToCompleteSuccess(status, "computeRadiativeTransfer: finished with photons")
    else
      call setStateToFailure(status, "computeRadiativeTransfer: Didn't process any photons.")
    end if
  end subroutine computeRadiativeTransferBatchMoments

  ! The batch moments of every process's share of a loop, summed over the processes: ONE all-reduce of ONE packed float64
  ! buffer (sums | sums of squares | number of batches) -- where the reference's drivers call sumAcrossProcesses ten times on real(4)
  ! fields (monteCarloDriver.f95:333-352 over Code/multipleProcesses_mpi.f95:57-131).  Afterwards reportBatchMoments reports
  ! the whole loop on every process.  A collective: every process calls it.
  subroutine sumBatchMomentsAcrossProcesses(thisIntegrator, status)
    type(integrator),   intent(inout) :: thisIntegrator
    type(ErrorMessage), intent(inout) :: status
    real(c_double), dimension(:), allocatable :: packed
    integer :: n

    if(.not. associated(thisIntegrator%momentSums)) then
      call setStateToFailure(status, "sumBatchMomentsAcrossProcesses: no batch moments have been computed.")
      return
    end if
    n = size(thisIntegrator%momentSums)
    allocate(packed(2 * n + 1))
    packed(1:n)         = thisIntegrator%momentSums
    packed(n + 1:2 * n) = thisIntegrator%momentSquares
    packed(2 * n + 1)   = real(thisIntegrator%momentBatches, c_double)
    packed = sumAcrossProcesses(packed)
    thisIntegrator%momentSums    = packed(1:n)
    thisIntegrator%momentSquares = packed(n + 1:2 * n)
    thisIntegrator%momentBatches = nint(packed(2 * n + 1))
    deallocate(packed)
  end subroutine sumBatchMomentsAcrossProcesses

  subroutine reportBatchMoments(thisIntegrator, numBatches, meanFluxUpStats, meanFluxDownStats, meanFluxAbsorbedStats,      &
                                fluxUpStats, fluxDownStats, fluxAbsorbedStats, absorbedProfileStats, volumeAbsorptionStats, &
                                meanIntensityStats, intensityStats, status)
    type(integrator),                      intent(in   ) :: thisIntegrator
    integer,                     optional, intent(  out) :: numBatches
    real, dimension(2),          optional, intent(  out) :: meanFluxUpStats, meanFluxDownStats, meanFluxAbsorbedStats
    real, dimension(:, :, :),    optional, intent(  out) :: fluxUpStats, fluxDownStats, fluxAbsorbedStats   ! (nx, ny, 2)
    real, dimension(:, :),       optional, intent(  out) :: absorbedProfileStats                            ! (nz, 2)
    real, dimension(:, :, :, :), optional, intent(  out) :: volumeAbsorptionStats                           ! (nx, ny, nz, 2)
    real, dimension(:, :),       optional, intent(  out) :: meanIntensityStats                              ! (nDir, 2)
    real, dimension(:, :, :, :), optional, intent(  out) :: intensityStats                                  ! (nx, ny, nDir, 2)
    type(ErrorMessage),                    intent(inout) :: status
    type(i3rc_moments_layout) :: layout
    integer :: nx, ny, nz, nDir, i, j, k
    integer(c_int64_t) :: at

    if(.not. associated(thisIntegrator%momentSums)) then
      call setStateToFailure(status, "reportBatchMoments: no batch moments have been computed.")
      return
    end if
    if(.not. deviceCall(thisIntegrator, i3rc_hip_get_moments_layout(thisIntegrator%device, layout), "reportBatchMoments", status)) return
    if(layout%total /= size(thisIntegrator%momentSums)) then
      call setStateToFailure(status, "reportBatchMoments: the problem has changed since the moments were computed.")
      return
    end if
    nx = size(thisIntegrator%fluxUp, 1); ny = size(thisIntegrator%fluxUp, 2); nz = size(thisIntegrator%volumeAbsorption, 3)
    nDir = 0
    if(thisIntegrator%computeIntensity) nDir = size(thisIntegrator%intensity, 3)
    if(present(numBatches)) numBatches = thisIntegrator%momentBatches
    if(present(meanFluxUpStats))       meanFluxUpStats       = scalarMoments(layout%meanFluxUp)
    if(present(meanFluxDownStats))     meanFluxDownStats     = scalarMoments(layout%meanFluxDown)
    if(present(meanFluxAbsorbedStats)) meanFluxAbsorbedStats = scalarMoments(layout%meanFluxAbsorbed)
    if(present(fluxUpStats))       call fieldMoments(layout%fluxUp,       nx * ny, fluxUpStats,       "fluxUpStats")
    if(present(fluxDownStats))     call fieldMoments(layout%fluxDown,     nx * ny, fluxDownStats,     "fluxDownStats")
    if(present(fluxAbsorbedStats)) call fieldMoments(layout%fluxAbsorbed, nx * ny, fluxAbsorbedStats, "fluxAbsorbedStats")
    if(present(absorbedProfileStats)) then
      if(any(shape(absorbedProfileStats) /= (/ nz, 2 /))) then
        call setStateToFailure(status, "reportBatchMoments: absorbedProfileStats array is the wrong size")
      else
        absorbedProfileStats(:, 1) = real(thisIntegrator%momentSums   (layout%absorbedProfile + 1:layout%absorbedProfile + nz))
        absorbedProfileStats(:, 2) = real(thisIntegrator%momentSquares(layout%absorbedProfile + 1:layout%absorbedProfile + nz))
      end if
    end if
    if(present(volumeAbsorptionStats)) then
      if(any(shape(volumeAbsorptionStats) /= (/ nx, ny, nz, 2 /))) then
        call setStateToFailure(status, "reportBatchMoments: volumeAbsorptionStats array is the wrong size")
      else
        ! (element by element: a cloud field's volume is millions of cells, and array-valued temporaries of that size have no place on a stack)
        do k = 1, nz
          do j = 1, ny
            do i = 1, nx
              at = layout%volumeAbsorption + i + nx * ((j - 1) + ny * (k - 1))
              volumeAbsorptionStats(i, j, k, 1) = real(thisIntegrator%momentSums(at))
              volumeAbsorptionStats(i, j, k, 2) = real(thisIntegrator%momentSquares(at))
            end do
          end do
        end do
      end if
    end if
    if(present(meanIntensityStats)) then
      if(nDir < 1) then
        call setStateToFailure(status, "reportBatchMoments: intensity information not available")
      else if(any(shape(meanIntensityStats) /= (/ nDir, 2 /))) then
        call setStateToFailure(status, "reportBatchMoments: requesting mean intensity in the wrong number of directions.")
      else
        meanIntensityStats(:, 1) = real(thisIntegrator%momentSums   (layout%meanIntensity + 1:layout%meanIntensity + nDir))
        meanIntensityStats(:, 2) = real(thisIntegrator%momentSquares(layout%meanIntensity + 1:layout%meanIntensity + nDir))
      end if
    end if
    if(present(intensityStats)) then
      if(nDir < 1) then
        call setStateToFailure(status, "reportBatchMoments: intensity information not available")
      else if(any(shape(intensityStats) /= (/ nx, ny, nDir, 2 /))) then
        call setStateToFailure(status, "reportBatchMoments: intensity array has wrong dimensions.")
      else
        do k = 1, nDir
          do j = 1, ny
            do i = 1, nx
              at = layout%intensity + i + nx * ((j - 1) + ny * (k - 1))
              intensityStats(i, j, k, 1) = real(thisIntegrator%momentSums(at))
              intensityStats(i, j, k, 2) = real(thisIntegrator%momentSquares(at))
            end do
          end do
        end do
      end if
    end if
    if(.not. stateIsFailure(status)) call setStateToSuccess(status)
  contains
    function scalarMoments(at) result(m)
      integer(c_int64_t), intent(in) :: at
      real :: m(2)
      m(1) = real(thisIntegrator%momentSums(at + 1)); m(2) = real(thisIntegrator%momentSquares(at + 1))
    end function scalarMoments
    subroutine fieldMoments(at, n, to, name)
      integer(c_int64_t),       intent(in ) :: at
      integer,                  intent(in ) :: n
      real, dimension(:, :, :), intent(out) :: to
      character(len = *),       intent(in ) :: name
      integer :: ii, jj
      if(any(shape(to) /= (/ nx, ny, 2 /))) then
        call setStateToFailure(status, "reportBatchMoments: " // name // " array is the wrong size")
      else
        do jj = 1, ny
          do ii = 1, nx
            to(ii, jj, 1) = real(thisIntegrator%momentSums   (at + ii + nx * (jj - 1)))
            to(ii, jj, 2) = real(thisIntegrator%momentSquares(at + ii + nx * (jj - 1)))
          end do
        end do
      end if
    end subroutine fieldMoments
  end subroutine reportBatchMoments

  ! ------------------------------------------------------------------------------------------------
  ! Reporting
  ! ------------------------------------------------------------------------------------------------
  subroutine reportResults(thisIntegrator, meanFluxUp, meanFluxDown, meanFluxAbsorbed, fluxUp, fluxDown, fluxAbsorbed, &
                           absorbedProfile, volumeAbsorption, meanIntensity, intensity, status)
    type(integrator),                   intent(in   ) :: thisIntegrator
    real,                     optional, intent(  out) :: meanFluxUp, meanFluxDown, meanFluxAbsorbed
    real, dimension(:, :),    optional, intent(  out) :: fluxUp, fluxDown, fluxAbsorbed
    real, dimension(:),       optional, intent(  out) :: absorbedProfile
    real, dimension(:, :, :), optional, intent(  out) :: volumeAbsorption
    real, dimension(:),       optional, intent(  out) :: meanIntensity
    real, dimension(:, :, :), optional, intent(  out) :: intensity
    type(ErrorMessage),                 intent(inout) :: status
    integer :: nColumns, d

    if(.not. associated(thisIntegrator%fluxUp)) then
      call setStateToFailure(status, "reportResults: integrator hasn't been initialized.")
      return
    end if
    if(.not. thisIntegrator%resultsValid .and. (associated(thisIntegrator%batchTallies) .or. associated(thisIntegrator%momentSums))) then
      ! a streamed loop has been announced (computeRadiativeTransferBatches) and none of its batches selected yet, or the last
      ! call gathered batch moments (computeRadiativeTransferBatchMoments: reportBatchMoments reports those): no single batch is
      ! current, and an earlier call's numbers are not this call's
      call setStateToFailure(status, "reportResults: no batch is current (selectBatchResults / reportBatchMoments).")
      return
    end if
    nColumns = size(thisIntegrator%fluxUp)
    if(present(meanFluxUp))       meanFluxUp       = sum(thisIntegrator%fluxUp)       / nColumns
    if(present(meanFluxDown))     meanFluxDown     = sum(thisIntegrator%fluxDown)     / nColumns
    if(present(meanFluxAbsorbed)) meanFluxAbsorbed = sum(thisIntegrator%fluxAbsorbed) / nColumns
    if(present(fluxUp))       call copyField(thisIntegrator%fluxUp,       fluxUp,       "fluxUp")
    if(present(fluxDown))     call copyField(thisIntegrator%fluxDown,     fluxDown,     "fluxDown")
    if(present(fluxAbsorbed)) call copyField(thisIntegrator%fluxAbsorbed, fluxAbsorbed, "fluxAbsorbed")
    if(present(absorbedProfile)) then
      if(size(absorbedProfile) /= size(thisIntegrator%volumeAbsorption, 3)) then
        call setStateToFailure(status, "reportResults: absorbedProfile array is the wrong size")
      else
        absorbedProfile(:) = sum(sum(thisIntegrator%volumeAbsorption, dim = 1), dim = 1) / nColumns
      end if
    end if
    if(present(volumeAbsorption)) then
      if(any(shape(volumeAbsorption) /= shape(thisIntegrator%volumeAbsorption))) then
        call setStateToFailure(status, "reportResults: volumeAbsorption array is the wrong size")
      else
        volumeAbsorption = thisIntegrator%volumeAbsorption
      end if
    end if
    if(present(meanIntensity)) then
      if(.not. associated(thisIntegrator%intensity)) then
        call setStateToFailure(status, "reportResults: intensity information not available")
      else if(size(thisIntegrator%intensity, 3) /= size(meanIntensity)) then
        call setStateToFailure(status, "reportResults: requesting mean intensity in the wrong number of directions.")
      else
        do d = 1, size(meanIntensity)
          meanIntensity(d) = sum(thisIntegrator%intensity(:, :, d)) / nColumns
        end do
      end if
    end if
    if(present(intensity)) then
      if(.not. associated(thisIntegrator%intensity)) then
        call setStateToFailure(status, "reportResults: intensity information not available")
      else if(any(shape(intensity) /= shape(thisIntegrator%intensity))) then
        call setStateToFailure(status, "reportResults: intensity array has wrong dimensions.")
      else
        intensity = thisIntegrator%intensity
      end if
    end if
    if(.not. stateIsFailure(status)) call setStateToSuccess(status)
  contains
    subroutine copyField(from, to, name)
      real, dimension(:, :), intent(in ) :: from
      real, dimension(:, :), intent(out) :: to
      character(len = *),    intent(in ) :: name
      if(any(shape(to) /= shape(from))) then
        call setStateToFailure(status, "reportResults: " // name // " array is the wrong size")
      else
        to = from
      end if
    end subroutine copyField
  end subroutine reportResults

  ! ------------------------------------------------------------------------------------------------
  ! Copy / finalize
  ! ------------------------------------------------------------------------------------------------
  function copy_Integrator(original) result(copy)
    type(integrator), intent(in) :: original
    type(integrator)             :: copy
    type(ErrorMessage) :: status
    integer :: c, nx, ny, nz, nc

    if(.not. associated(original%totalExt)) return
    nx = size(original%totalExt, 1); ny = size(original%totalExt, 2); nz = size(original%totalExt, 3)
    nc = size(original%cumulativeExt, 4)
    allocate(copy%xPosition(nx + 1), copy%yPosition(ny + 1), copy%zPosition(nz + 1), copy%totalExt(nx, ny, nz), &
             copy%cumulativeExt(nx, ny, nz, nc), copy%ssa(nx, ny, nz, nc), copy%phaseFunctionIndex(nx, ny, nz, nc), &
             copy%forwardTables(nc))
    copy%xPosition = original%xPosition; copy%yPosition = original%yPosition; copy%zPosition = original%zPosition
    copy%totalExt = original%totalExt; copy%cumulativeExt = original%cumulativeExt; copy%ssa = original%ssa
    copy%phaseFunctionIndex = original%phaseFunctionIndex
    do c = 1, nc
      copy%forwardTables(c) = copy_PhaseFunctionTable(original%forwardTables(c))
    end do
    copy%minForwardTableSize = original%minForwardTableSize
    copy%minInverseTableSize = original%minInverseTableSize
    copy%parameters          = original%parameters
    copy%hybridPhaseFunWidth = original%hybridPhaseFunWidth
    copy%deviceIndex         = original%deviceIndex
    call createDeviceInstance(copy, status)
    if(stateIsFailure(status)) return
    allocate(copy%fluxUp(nx, ny), copy%fluxDown(nx, ny), copy%fluxAbsorbed(nx, ny), copy%volumeAbsorption(nx, ny, nz))
    copy%fluxUp = original%fluxUp; copy%fluxDown = original%fluxDown; copy%fluxAbsorbed = original%fluxAbsorbed
    copy%volumeAbsorption = original%volumeAbsorption
    copy%readyToCompute = original%readyToCompute
    copy%resultsValid   = original%resultsValid
    if(original%useSurfaceBDRF) then
      call specifyParameters(copy, surfaceBDRF = original%surfaceBDRF, status = status)
    else
      call specifyParameters(copy, surfaceAlbedo = original%parameters%surfaceAlbedo, status = status)
    end if
    if(associated(original%intensityDirections)) then
      ! directions are stored as cosines: recover (mu, phi in degrees) for the public setter
      call specifyParameters(copy, intensityMus = original%intensityDirections(3, :),                                  &
                             intensityPhis = modulo(atan2(original%intensityDirections(2, :),                          &
                                                          original%intensityDirections(1, :)) * 180. / Pi, 360.),      &
                             status = status)
      copy%intensity = original%intensity
      copy%intensityByComponent = original%intensityByComponent
    end if
  end function copy_Integrator

  subroutine finalize_Integrator(thisIntegrator)
    type(integrator), intent(inout) :: thisIntegrator
    integer :: c, rc
    if(c_associated(thisIntegrator%device)) rc = i3rc_hip_destroy(thisIntegrator%device)
    thisIntegrator%device = c_null_ptr
    if(associated(thisIntegrator%xPosition))           deallocate(thisIntegrator%xPosition)
    if(associated(thisIntegrator%yPosition))           deallocate(thisIntegrator%yPosition)
    if(associated(thisIntegrator%zPosition))           deallocate(thisIntegrator%zPosition)
    if(associated(thisIntegrator%totalExt))            deallocate(thisIntegrator%totalExt)
    if(associated(thisIntegrator%cumulativeExt))       deallocate(thisIntegrator%cumulativeExt)
    if(associated(thisIntegrator%ssa))                 deallocate(thisIntegrator%ssa)
    if(associated(thisIntegrator%phaseFunctionIndex))  deallocate(thisIntegrator%phaseFunctionIndex)
    if(associated(thisIntegrator%forwardTables)) then
      do c = 1, size(thisIntegrator%forwardTables)
        call finalize_PhaseFunctionTable(thisIntegrator%forwardTables(c))
      end do
      deallocate(thisIntegrator%forwardTables)
    end if
    if(associated(thisIntegrator%inverseSizeOnDevice)) deallocate(thisIntegrator%inverseSizeOnDevice)
    if(associated(thisIntegrator%forwardSizeOnDevice)) deallocate(thisIntegrator%forwardSizeOnDevice)
    call finalize_surfaceDescription(thisIntegrator%surfaceBDRF)
    if(associated(thisIntegrator%intensityDirections))  deallocate(thisIntegrator%intensityDirections)
    if(associated(thisIntegrator%fluxUp))               deallocate(thisIntegrator%fluxUp)
    if(associated(thisIntegrator%fluxDown))             deallocate(thisIntegrator%fluxDown)
    if(associated(thisIntegrator%fluxAbsorbed))         deallocate(thisIntegrator%fluxAbsorbed)
    if(associated(thisIntegrator%volumeAbsorption))     deallocate(thisIntegrator%volumeAbsorption)
    if(associated(thisIntegrator%intensity))            deallocate(thisIntegrator%intensity)
    if(associated(thisIntegrator%intensityByComponent)) deallocate(thisIntegrator%intensityByComponent)
    if(associated(thisIntegrator%batchTallies))         deallocate(thisIntegrator%batchTallies)
    if(associated(thisIntegrator%momentSums))           deallocate(thisIntegrator%momentSums, thisIntegrator%momentSquares)
    thisIntegrator%readyToCompute = .false.; thisIntegrator%computeIntensity = .false.
    thisIntegrator%useSurfaceBDRF = .false.
  end subroutine finalize_Integrator
end module monteCarloRadiativeTransfer
