! Self test of the Fortran shell.  "cpu": everything that needs no GPU (status object, RNG known answers, tables,
! domain file round trip), prints key numbers for tests/test_fortran_shell.py.  "gpu": plane-parallel slab through
! new_Integrator / specifyParameters / computeRadiativeTransfer / reportResults on the device.
program shellSelfTest
  use ErrorMessages
  use RandomNumbers
  use numericUtilities
  use scatteringPhaseFunctions
  use inversePhaseFunctions
  use opticalProperties
  use surfaceProperties
  use monteCarloIllumination
  use monteCarloRadiativeTransfer
  implicit none
  character(len = 16) :: mode
  type(ErrorMessage) :: status

  mode = "cpu"
  if(command_argument_count() >= 1) call get_command_argument(1, mode)
  if(trim(mode) == "cpu") then
    call cpuChecks
  else
    call gpuChecks
  end if
contains
  subroutine cpuChecks
    type(randomNumberSequence) :: r
    type(phaseFunction)        :: hg
    type(phaseFunctionTable)   :: table
    type(domain)               :: d, d2
    real    :: inverse(10001, 1), forward(10001, 1), angles(10001)
    real    :: ext(3, 2, 4), ssa(3, 2, 4), total(3, 2, 4), cum(3, 2, 4, 2), alb(3, 2, 4, 2), total2(3, 2, 4), cum2(3, 2, 4, 2)
    integer :: idx(3, 2, 4), pfi(3, 2, 4, 2), i, nx, ny, nz, nc
    character(len = 64) :: names(2)

    r = new_RandomNumberSequence(100)
    print '(a, 5f12.9)', "mt_scalar ", (getRandomReal(r), i = 1, 5)
    r = new_RandomNumberSequence((/ 10, 1 /))
    print '(a, 5i12)',   "mt_vector ", (getRandomInt(r), i = 1, 5)
    hg = new_PhaseFunction(0.85**(/ (i, i = 1, 64) /), status = status)
    table = new_PhaseFunctionTable((/ hg /), key = (/ 1. /), status = status)
    call computeInversePhaseFuncTable(table, inverse, status)
    print '(a, 7f10.6)', "inverse   ", inverse(1, 1), inverse(2, 1), inverse(2501, 1), inverse(5001, 1), inverse(7501, 1), &
                         inverse(10000, 1), inverse(10001, 1)
    angles = (/ (i, i = 0, 10000) /) / real(10000) * 3.14159265358979312
    call getPhaseFunctionValues(table, angles, forward, status)
    print '(a, 3es16.8)', "forward   ", forward(1, 1), forward(5001, 1), forward(10001, 1)
    ! domain with a 3-D and a horizontally uniform component, written and read back
    call random_seed()
    do i = 1, 4
      ext(:, :, i) = 0.01 * i; ssa(:, :, i) = 0.9; idx(:, :, i) = 1
    end do
    ext(2, 1, 3) = 0.; idx(2, 1, 3) = 0; ssa(2, 1, 3) = 0.
    d = new_Domain((/ 0., 10., 20., 30. /), (/ 0., 5., 10. /), (/ 0., 1., 2., 3., 4. /), status)
    call addOpticalComponent(d, "cloud", ext, ssa, idx, table, status = status)
    call addOpticalComponent(d, "gas", (/ 0.001, 0.002 /), (/ 0.5, 0.5 /), (/ 1, 1 /), table, zLevelBase = 2, status = status)
    call getOpticalPropertiesByComponent(d, total, cum, alb, pfi, status = status)
    call write_Domain(d, "/tmp/i3rc_shell_selftest.dom", status)
    call read_Domain("/tmp/i3rc_shell_selftest.dom", d2, status)
    if(stateIsFailure(status)) print *, "domain round trip FAILED"
    call getInfo_Domain(d2, numX = nx, numY = ny, numZ = nz, numberOfComponents = nc, componentNames = names, status = status)
    call getOpticalPropertiesByComponent(d2, total2, cum2, alb, pfi, status = status)
    print '(a, 4i4, 1x, a, 1x, a)', "domain    ", nx, ny, nz, nc, trim(names(1)), trim(names(2))
    print '(a, 2es16.8, l2)', "roundtrip ", maxval(abs(total - total2)), maxval(abs(cum - cum2)), stateIsFailure(status)
    print '(a, 3f10.6)', "layers    ", total(1, 1, 1), total(1, 1, 2), cum(1, 1, 2, 1)
    call finalize_Domain(d); call finalize_Domain(d2)
    print '(a)', "cpu checks done"
  end subroutine cpuChecks

  subroutine gpuChecks
    type(domain)               :: slab
    type(integrator)           :: mc
    type(randomNumberSequence) :: randoms
    type(photonStream)         :: photons
    type(phaseFunction)        :: hg
    type(phaseFunctionTable)   :: table
    type(surfaceDescription)   :: surface
    integer, parameter :: nBatches = 4, nPhotons = 200000
    real    :: fluxUp(1, 1), fluxDown(1, 1), fluxAbs(1, 1), up(nBatches), down(nBatches), meanI(2)
    integer :: batch, i

    hg = new_PhaseFunction(0.85**(/ (i, i = 1, 64) /), status = status)
    table = new_PhaseFunctionTable((/ hg /), key = (/ 1. /), status = status)
    slab = new_Domain((/ 0., 500. /), (/ 0., 500. /), (/ 0., 250. /), status)
    call addOpticalComponent(slab, "cloud", reshape((/ 1. / 250. /), (/ 1, 1, 1 /)), reshape((/ 1. /), (/ 1, 1, 1 /)), &
                             reshape((/ 1 /), (/ 1, 1, 1 /)), table, status = status)
    mc = new_Integrator(slab, status)
    if(stateIsFailure(status)) then
      print *, "new_Integrator failed"; stop 1
    end if
    call specifyParameters(mc, surfaceAlbedo = 0., status = status)
    do batch = 1, nBatches
      randoms = new_RandomNumberSequence(seed = (/ batch, 10 /))
      photons = new_PhotonStream(0.5, 0., numberOfPhotons = nPhotons, randomNumbers = randoms, status = status)
      call computeRadiativeTransfer(mc, randoms, photons, status)
      call reportResults(mc, fluxUp = fluxUp, fluxDown = fluxDown, fluxAbsorbed = fluxAbs, status = status)
      up(batch) = fluxUp(1, 1); down(batch) = fluxDown(1, 1)
      call finalize_PhotonStream(photons)
    end do
    ! reference (planeParallel.nml as shipped, SURVEY.md 6): Fup 0.16420, Fdown 0.83580 +- 0.0036 at 4 x 1e4 photons
    print '(a, 2f10.5, l2)', "slab      ", sum(up) / nBatches, sum(down) / nBatches, morePhotonsExist(photons)
    ! Lambertian surface through the BRDF object + two radiance directions
    surface = new_SurfaceDescription((/ 0.3 /), status)
    call specifyParameters(mc, surfaceBDRF = surface, status = status)
    call specifyParameters(mc, intensityMus = (/ 1., 0.5 /), intensityPhis = (/ 0., 90. /), status = status)
    randoms = new_RandomNumberSequence(seed = (/ 1, 10 /))
    photons = new_PhotonStream(0.5, 0., numberOfPhotons = nPhotons, randomNumbers = randoms, status = status)
    call computeRadiativeTransfer(mc, randoms, photons, status)
    call reportResults(mc, fluxUp = fluxUp, fluxDown = fluxDown, meanIntensity = meanI, status = status)
    print '(a, 4f10.5, l2)', "surface   ", fluxUp(1, 1), fluxDown(1, 1), meanI, stateIsFailure(status)
    call finalize_Integrator(mc)
    call otherSources(slab)
    call streamedLoop(slab)
    print '(a)', "gpu checks done"
  end subroutine gpuChecks

  ! The batch loop as the shell's own entry points run it: announced and streamed (computeRadiativeTransferBatches /
  ! selectBatchResults), and with its statistics gathered on the device (computeRadiativeTransferBatchMoments / reportBatchMoments).
  ! Prints "streamed" with: reportResults before any batch was selected fails (T); sum over the batches of fluxUp taken batch by
  ! batch; the same from the device's moments; sum of squares likewise, twice; a change of the radiance directions in mid-loop
  ! makes the next selectBatchResults fail (T) instead of writing a block of another length into the loop's storage.
  subroutine streamedLoop(slab)
    type(domain), intent(in) :: slab
    type(integrator) :: mc
    integer, parameter :: nBatches = 12, nPhotons = 50000
    real    :: fluxUp(1, 1), upStats(1, 1, 2), meanUp(2)
    double precision :: s1, s2
    logical :: staleRefused, changeRefused
    integer :: batch, nDone

    mc = new_Integrator(slab, status)
    call specifyParameters(mc, surfaceAlbedo = 0.2, status = status)
    call computeRadiativeTransferBatches(mc, 7, 1, nBatches, 0.5, 0., nPhotons, status)
    if(stateIsFailure(status)) then
      print *, "computeRadiativeTransferBatches failed"; stop 1
    end if
    call reportResults(mc, fluxUp = fluxUp, status = status)        ! nothing selected yet: must not report an earlier call's numbers
    staleRefused = stateIsFailure(status)
    call setStateToSuccess(status)
    s1 = 0.d0; s2 = 0.d0
    do batch = 1, nBatches
      call selectBatchResults(mc, batch, status)
      call reportResults(mc, fluxUp = fluxUp, status = status)
      if(stateIsFailure(status)) then
        print *, "streamed batch failed ", batch; stop 1
      end if
      s1 = s1 + fluxUp(1, 1); s2 = s2 + dble(fluxUp(1, 1))**2
    end do
    call computeRadiativeTransferBatchMoments(mc, 7, 1, nBatches, 0.5, 0., nPhotons, status)
    call reportBatchMoments(mc, numBatches = nDone, meanFluxUpStats = meanUp, fluxUpStats = upStats, status = status)
    if(stateIsFailure(status) .or. nDone /= nBatches) then
      print *, "batch moments failed"; stop 1
    end if
    ! a parameter change in mid-loop: announced with no radiance directions, asked for with one
    call computeRadiativeTransferBatches(mc, 7, 1, nBatches, 0.5, 0., nPhotons, status)
    call selectBatchResults(mc, 2, status)
    call specifyParameters(mc, intensityMus = (/ 1. /), intensityPhis = (/ 0. /), status = status)
    call setStateToSuccess(status)
    call selectBatchResults(mc, 5, status)
    changeRefused = stateIsFailure(status)
    call setStateToSuccess(status)
    print '(a, l2, 2f12.6, 2f12.7, 2f12.6, l2)', "streamed  ", staleRefused, s1, upStats(1, 1, 1), s2, upStats(1, 1, 2), meanUp, changeRefused
    call finalize_Integrator(mc)
  end subroutine streamedLoop

  ! The photon sources that hand the kernel explicit start positions and directions (monteCarloIllumination.f95
  ! :106-424): on a horizontally uniform slab the azimuth and the place of entry do not matter, so RandomAzimuth and
  ! Spotlight must give the Directional fluxes; Flux and the internal (detector) sources must conserve energy.
  subroutine otherSources(slab)
    type(domain), intent(in) :: slab
    type(integrator)           :: mc
    type(randomNumberSequence) :: randoms
    type(photonStream)         :: photons
    integer, parameter :: nPhotons = 400000
    real    :: fluxUp(1, 1), fluxDown(1, 1), results(2, 4)
    integer :: kind

    mc = new_Integrator(slab, status)
    call specifyParameters(mc, surfaceAlbedo = 0., status = status)
    do kind = 1, 4
      randoms = new_RandomNumberSequence(seed = (/ 20 + kind, 3 /))
      select case(kind)
        case(1); photons = new_PhotonStream(0.5, numberOfPhotons = nPhotons, randomNumbers = randoms, status = status)
        case(2); photons = new_PhotonStream(0.5, 0., 0.2, 0.4, numberOfPhotons = nPhotons, randomNumbers = randoms, status = status)
        case(3); photons = new_PhotonStream(numberOfPhotons = nPhotons, randomNumbers = randoms, status = status)
        case(4); photons = new_PhotonStream(0.5, 0.5, 0.5, .true., numberOfPhotons = nPhotons, randomNumbers = randoms, &
                                            status = status)
      end select
      if(stateIsFailure(status)) then
        print *, "photon source ", kind, " failed"; stop 1
      end if
      call computeRadiativeTransfer(mc, randoms, photons, status)
      if(stateIsFailure(status)) then
        print *, "computeRadiativeTransfer failed for photon source ", kind; stop 1
      end if
      call reportResults(mc, fluxUp = fluxUp, fluxDown = fluxDown, status = status)
      results(:, kind) = (/ fluxUp(1, 1), fluxDown(1, 1) /)
      call finalize_PhotonStream(photons)
    end do
    print '(a, 8f10.5)', "sources   ", results
    call finalize_Integrator(mc)
  end subroutine otherSources
end program shellSelfTest
