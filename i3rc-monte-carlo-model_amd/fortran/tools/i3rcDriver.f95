! i3rcDriver -- the shell's own production driver for the MI355X integrator.
!
! Same interface as the reference's Example-Drivers/monteCarloDriver.f95 so that existing run decks keep working:
! the five namelists radiativeTransfer / monteCarlo / algorithms / output / fileNames (:90-103), the batch and seed
! scheme (seed = (/iseed, batch/), batches split over processes :264-326), mean and standard error from the first two
! moments over batches (:358-378, including its scaling by solarFlux), and the ASCII result files (:484-604; same
! header keys and line formats, which downstream scripts parse).  The unchanged reference driver also links against
! the shell (make linkcheck); this one exists so that a tree without the reference can run end to end.
!   i3rcDriver namelistFile
program i3rcDriver
  use ErrorMessages,               only: ErrorMessage, stateIsFailure
  use MultipleProcesses,           only: MasterProc, sumAcrossProcesses, initializeProcesses, finalizeProcesses, &
                                         synchronizeProcesses
  use RandomNumbers,               only: randomNumberSequence, new_RandomNumberSequence, finalize_RandomNumberSequence
  use opticalProperties,           only: domain, read_Domain, getInfo_Domain, finalize_Domain
  use monteCarloIllumination,      only: photonStream, new_PhotonStream, finalize_PhotonStream
  use monteCarloRadiativeTransfer, only: integrator, new_Integrator, specifyParameters, isReady_Integrator, &
                                         finalize_Integrator, computeRadiativeTransfer, reportResults,           &
                                         computeRadiativeTransferBatches, selectBatchResults, &
                                         computeRadiativeTransferBatchMoments, reportBatchMoments, sumBatchMomentsAcrossProcesses
  use UserInterface,               only: printStatus, getOneArgument
  implicit none

  integer, parameter :: maxDirections = 20
  ! -- namelist variables (names and defaults as in the reference driver :58-87)
  real    :: solarFlux = 1., surfaceAlbedo = 0., solarMu = 1., solarAzimuth = 0.
  real    :: intensityMus(maxDirections) = 0., intensityPhis(maxDirections) = 0.
  integer :: numPhotonsPerBatch = 0, numBatches = 100, iseed = 10, nPhaseIntervals = 10001
  logical :: useRayTracing = .true., useRussianRoulette = .true.
  logical :: useHybridPhaseFunsForIntenCalcs = .false.
  real    :: hybridPhaseFunWidth = 7.
  integer :: numOrdersOrigPhaseFunIntenCalcs = 0
  logical :: useRussianRouletteForIntensity = .true.
  real    :: zetaMin = 0.3
  logical :: limitIntensityContributions = .false.
  real    :: maxIntensityContribution = 77.
  logical :: reportVolumeAbsorption = .false., reportAbsorptionProfile = .false.
  character(len = 256) :: domainFileName = "", outputFluxFile = "", outputRadFile = "", outputAbsProfFile = "", &
                          outputAbsVolumeFile = "", outputNetcdfFile = ""
  namelist /radiativeTransfer/ solarFlux, solarMu, solarAzimuth, surfaceAlbedo, intensityMus, intensityPhis
  namelist /monteCarlo/ numPhotonsPerBatch, numBatches, iseed, nPhaseIntervals
  namelist /algorithms/ useRayTracing, useRussianRoulette, useHybridPhaseFunsForIntenCalcs, hybridPhaseFunWidth, &
                        numOrdersOrigPhaseFunIntenCalcs, useRussianRouletteForIntensity, zetaMin,               &
                        limitIntensityContributions, maxIntensityContribution
  namelist /output/ reportVolumeAbsorption, reportAbsorptionProfile
  namelist /fileNames/ domainFileName, outputRadFile, outputFluxFile, outputAbsProfFile, outputAbsVolumeFile, outputNetcdfFile

  character(len = 256) :: namelistFile
  integer :: nx, ny, nz, nDir, numProcs, thisProc, perProc, batch, firstBatch
  integer :: inFlight, groupSize, groupStart, inGroup, deviceMoments
  real    :: tallyWords
  character(len = 32) :: envText
  logical :: wantRadiance, momentsAreGlobal = .false.
  real    :: t0, t1, t2, cpuSetup
  integer :: nc, v, rc                       ! netCDF result file: file id, variable id, return code
  real, allocatable :: xEdges(:), yEdges(:), zEdges(:)
  real, allocatable :: up(:, :), down(:, :), absorbed(:, :), profile(:), volume(:, :, :), radiance(:, :, :)
  ! first and second moments over batches, one slab per quantity
  real, allocatable :: mUp(:, :, :), mDown(:, :, :), mAbs(:, :, :), mProfile(:, :), mVolume(:, :, :, :), mRad(:, :, :, :)
  real :: meanUp, meanDown, meanAbs, mMeans(3, 2)
  type(domain)               :: cloud
  type(ErrorMessage)         :: status
  type(randomNumberSequence) :: randoms
  type(photonStream)         :: photons
  type(integrator)           :: mc

  call initializeProcesses(numProcs, thisProc)
  call cpu_time(t0)
  namelistFile = getOneArgument()
  open(unit = 11, file = trim(namelistFile), status = "old", action = "read")
  read(11, nml = radiativeTransfer); rewind(11)
  read(11, nml = monteCarlo);        rewind(11)
  read(11, nml = algorithms);        rewind(11)
  read(11, nml = output);            rewind(11)
  read(11, nml = fileNames)
  close(11)
  nDir = count(abs(intensityMus) > 0.)
  wantRadiance = nDir > 0 .and. (len_trim(outputRadFile) > 0 .or. len_trim(outputNetcdfFile) > 0)
  if(.not. wantRadiance) outputRadFile = ""

  ! -- problem set-up
  call read_Domain(domainFileName, cloud, status);                                call printStatus(status)
  call getInfo_Domain(cloud, numX = nx, numY = ny, numZ = nz, status = status)
  allocate(xEdges(nx + 1), yEdges(ny + 1), zEdges(nz + 1))
  call getInfo_Domain(cloud, xPosition = xEdges, yPosition = yEdges, zPosition = zEdges, status = status)
  mc = new_Integrator(cloud, status);                                             call printStatus(status)
  call finalize_Domain(cloud)
  call specifyParameters(mc, surfaceAlbedo = surfaceAlbedo, minInverseTableSize = nPhaseIntervals, status = status)
  call printStatus(status)
  if(wantRadiance) then
    call specifyParameters(mc, minForwardTableSize = nPhaseIntervals, intensityMus = intensityMus(:nDir), &
                           intensityPhis = intensityPhis(:nDir), computeIntensity = .true., status = status)
    call printStatus(status)
    call specifyParameters(mc, useHybridPhaseFunsForIntenCalcs = useHybridPhaseFunsForIntenCalcs,        &
                           hybridPhaseFunWidth = hybridPhaseFunWidth,                                    &
                           numOrdersOrigPhaseFunIntenCalcs = numOrdersOrigPhaseFunIntenCalcs,            &
                           useRussianRouletteForIntensity = useRussianRouletteForIntensity, zetaMin = zetaMin, &
                           limitIntensityContributions = limitIntensityContributions,                    &
                           maxIntensityContribution = maxIntensityContribution, status = status)
    call printStatus(status)
  end if
  call specifyParameters(mc, useRayTracing = useRayTracing, useRussianRoulette = useRussianRoulette, status = status)
  call printStatus(status)
  if(.not. isReady_Integrator(mc)) stop "Integrator is not ready."

  allocate(up(nx, ny), down(nx, ny), absorbed(nx, ny), profile(nz), volume(nx, ny, nz))
  allocate(mUp(nx, ny, 2), mDown(nx, ny, 2), mAbs(nx, ny, 2), mProfile(nz, 2), mVolume(nx, ny, nz, 2))
  mUp = 0.; mDown = 0.; mAbs = 0.; mProfile = 0.; mVolume = 0.; mMeans = 0.
  if(wantRadiance) then
    allocate(radiance(nx, ny, nDir), mRad(nx, ny, nDir, 2))
    mRad = 0.
  end if

  ! one-photon run: builds the phase-function tables and proves the set-up before the batches start
  randoms = new_RandomNumberSequence(seed = (/ iseed, 0 /))
  photons = new_PhotonStream(solarMu, solarAzimuth, numberOfPhotons = 1, randomNumbers = randoms, status = status)
  call computeRadiativeTransfer(mc, randoms, photons, status);                    call printStatus(status)
  call finalize_PhotonStream(photons)
  call cpu_time(t1)
  call synchronizeProcesses
  t1 = sumAcrossProcesses(t1 - t0)          ! a collective: every rank takes part
  cpuSetup = t1
  if(MasterProc) print *, "Setup CPU time (secs, approx): ", int(t1)

  ! -- batches: the unit of work and of the error estimate
  numBatches = max(numBatches, 2)
  perProc = (numBatches + numProcs - 1) / numProcs
  numBatches = perProc * numProcs
  if(MasterProc) print *, "Doing ", perProc, " batches on each of ", numProcs, " processors."
  firstBatch = thisProc * perProc + 1
  ! The reference's loop -- per batch a sequence seeded (/ iseed, batch /), a photon stream, computeRadiativeTransfer,
  ! reportResults (monteCarloDriver.f95:283-326) -- runs here as computeRadiativeTransferBatches over groups of batches:
  ! the same photons batch by batch, but several batches share the GPU at a time, so that the long tail of one batch
  ! (a few photons with a thousand scatterings) is covered by the next -- flux problems of the common class even share ONE
  ! kernel launch per hundred batches or so (i3rc_hip_run_batches).  A group holds at most 4096 batches and 256 MB of raw
  ! tallies; I3RC_BATCHES_IN_FLIGHT=1 (and I3RC_FUSED=0) in the environment runs the batches one after the other.
  inFlight = 0
  call get_environment_variable("I3RC_BATCHES_IN_FLIGHT", envText, status = rc)
  if(rc == 0 .and. len_trim(envText) > 0) read(envText, *, iostat = rc) inFlight
  ! All this driver keeps of a batch are the first two moments of its results (below; monteCarloDriver.f95:300-321): by default they
  ! are gathered on the device (computeRadiativeTransferBatchMoments) and the batches' tallies -- megabytes each on a cloud
  ! field -- never come to the host.  I3RC_DRIVER_MOMENTS=0 in the environment keeps the batch-by-batch loop.
  deviceMoments = 1
  call get_environment_variable("I3RC_DRIVER_MOMENTS", envText, status = rc)
  if(rc == 0 .and. len_trim(envText) > 0) read(envText, *, iostat = rc) deviceMoments
  if(deviceMoments /= 0 .and. inFlight /= 1) then
    call computeRadiativeTransferBatchMoments(mc, iseed, firstBatch, perProc, solarMu, solarAzimuth, numPhotonsPerBatch, status)
    call printStatus(status)
    ! the processes' shares of the loop: ONE all-reduce of the packed float64 moments (where the reference's driver reduces ten
    ! real(4) fields one by one, :333-352) -- what reportBatchMoments hands out below is the whole loop's, on every process
    call sumBatchMomentsAcrossProcesses(mc, status)
    call printStatus(status)
    momentsAreGlobal = .true.
    call reportBatchMoments(mc, meanFluxUpStats = mMeans(1, :), meanFluxDownStats = mMeans(2, :), meanFluxAbsorbedStats = mMeans(3, :), &
                            fluxUpStats = mUp, fluxDownStats = mDown, fluxAbsorbedStats = mAbs, absorbedProfileStats = mProfile,        &
                            volumeAbsorptionStats = mVolume, status = status)
    call printStatus(status)
    if(wantRadiance) then
      call reportBatchMoments(mc, intensityStats = mRad, status = status)
      call printStatus(status)
    end if
    perProc = 0   ! (the loop below has nothing left to do)
  end if
  tallyWords = 3. * nx * ny + real(nx) * ny * nz + 2. * nDir * nx * ny
  groupSize = max(1, min(4096, int(256. * 1024. * 1024. / (8. * tallyWords))))
  do groupStart = firstBatch, firstBatch + perProc - 1, groupSize
    inGroup = min(groupSize, firstBatch + perProc - groupStart)
    call computeRadiativeTransferBatches(mc, iseed, groupStart, inGroup, solarMu, solarAzimuth, numPhotonsPerBatch, status, &
                                         batchesInFlight = inFlight)
    call printStatus(status)
    do batch = 1, inGroup
      call selectBatchResults(mc, batch, status)
      call reportResults(mc, meanUp, meanDown, meanAbs, up, down, absorbed, profile, volume, status = status)
      call accumulate0(mMeans(1, :), meanUp); call accumulate0(mMeans(2, :), meanDown); call accumulate0(mMeans(3, :), meanAbs)
      mUp(:, :, 1)   = mUp(:, :, 1)   + up;       mUp(:, :, 2)   = mUp(:, :, 2)   + up**2
      mDown(:, :, 1) = mDown(:, :, 1) + down;     mDown(:, :, 2) = mDown(:, :, 2) + down**2
      mAbs(:, :, 1)  = mAbs(:, :, 1)  + absorbed; mAbs(:, :, 2)  = mAbs(:, :, 2)  + absorbed**2
      mProfile(:, 1) = mProfile(:, 1) + profile;  mProfile(:, 2) = mProfile(:, 2) + profile**2
      mVolume(:, :, :, 1) = mVolume(:, :, :, 1) + volume; mVolume(:, :, :, 2) = mVolume(:, :, :, 2) + volume**2
      if(wantRadiance) then
        call reportResults(mc, intensity = radiance, status = status)
        mRad(:, :, :, 1) = mRad(:, :, :, 1) + radiance; mRad(:, :, :, 2) = mRad(:, :, :, 2) + radiance**2
      end if
      call printStatus(status)
    end do
  end do

  ! -- gather over processes, then mean and standard error from the two moments
  if(.not. momentsAreGlobal) then   ! (the batch-by-batch loop: real(4) sums, field by field, as the reference's driver reduces them)
    mMeans(1, :) = sumAcrossProcesses(mMeans(1, :)); mMeans(2, :) = sumAcrossProcesses(mMeans(2, :))
    mMeans(3, :) = sumAcrossProcesses(mMeans(3, :))
    mUp = sumAcrossProcesses(mUp); mDown = sumAcrossProcesses(mDown); mAbs = sumAcrossProcesses(mAbs)
    mProfile = sumAcrossProcesses(mProfile); mVolume = sumAcrossProcesses(mVolume)
    if(wantRadiance) mRad = sumAcrossProcesses(mRad)
  end if
  call synchronizeProcesses
  call cpu_time(t2)
  t2 = sumAcrossProcesses(t2 - t0)
  call finalizeProcesses
  if(MasterProc) print *, "Total CPU time (secs, approx): ", int(t2)

  call momentsToStatistics(mMeans(1, :)); call momentsToStatistics(mMeans(2, :)); call momentsToStatistics(mMeans(3, :))
  call reduce3(mUp); call reduce3(mDown); call reduce3(mAbs)
  mProfile = solarFlux * mProfile / numBatches
  mProfile(:, 2) = sqrt(max(0., mProfile(:, 2) - mProfile(:, 1)**2) / (numBatches - 1))
  mVolume = solarFlux * mVolume / numBatches
  mVolume(:, :, :, 2) = sqrt(max(0., mVolume(:, :, :, 2) - mVolume(:, :, :, 1)**2) / (numBatches - 1))
  if(wantRadiance) then
    mRad = solarFlux * mRad / numBatches
    mRad(:, :, :, 2) = sqrt(max(0., mRad(:, :, :, 2) - mRad(:, :, :, 1)**2) / (numBatches - 1))
  end if

  if(MasterProc) then
    if(len_trim(outputFluxFile) > 0)    call writeFluxFile
    if(len_trim(outputAbsProfFile) > 0) call writeProfileFile
    if(len_trim(outputAbsVolumeFile) > 0) call writeVolumeFile
    if(len_trim(outputRadFile) > 0)     call writeRadianceFile
    if(len_trim(outputFluxFile) + len_trim(outputAbsProfFile) + len_trim(outputAbsVolumeFile) + len_trim(outputRadFile) > 0) &
      print *, "Wrote ASCII results"
    if(len_trim(outputNetcdfFile) > 0) then
      call writeNetcdfFile
      print *, "Wrote netCDF results"
    end if
  end if
  call finalize_Integrator(mc)
contains
  subroutine accumulate0(m, x)
    real, intent(inout) :: m(2)
    real, intent(in   ) :: x
    m(1) = m(1) + x; m(2) = m(2) + x**2
  end subroutine accumulate0

  subroutine momentsToStatistics(m)
    real, intent(inout) :: m(2)
    m = solarFlux * m / numBatches
    m(2) = sqrt(max(0., m(2) - m(1)**2) / (numBatches - 1))
  end subroutine momentsToStatistics

  subroutine reduce3(m)
    real, intent(inout) :: m(:, :, :)
    m = solarFlux * m / numBatches
    m(:, :, 2) = sqrt(max(0., m(:, :, 2) - m(:, :, 1)**2) / (numBatches - 1))
  end subroutine reduce3

  subroutine writeHeader(unit, title, outputType)
    integer,            intent(in) :: unit
    character(len = *), intent(in) :: title, outputType
    write(unit, '(A)')                  '!   I3RC Monte Carlo 3D Solar Radiative Transfer: ' // title
    write(unit, '(A,A60)')              '!  Property_File=', domainFileName
    write(unit, '(A,I10)')              '!  Num_Photons=', numPhotonsPerBatch * numBatches
    write(unit, '(A,L1,A,L1)')          '!  PhotonTracing=', useRayTracing, '    Russian_Roulette=', useRussianRoulette
    write(unit, '(A,L1,A,F5.2)')        '!  Hybrid_Phase_Func_for_Radiance=', useHybridPhaseFunsForIntenCalcs, &
                                        '   Gaussian_Phase_Func_Width_deg=', hybridPhaseFunWidth
    if(outputType == "Pixel Radiance") then
      write(unit, '(A,L1,A,F5.2)')      '!  Intensity_uses_Russian_Roulette=', useRussianRouletteForIntensity, &
                                        '   Intensity_Russian_Roulette_zeta_min=', zetaMin
      write(unit, '(A,L1,A,F5.2)')      '!  limited_intensity_contributions=', limitIntensityContributions, &
                                        '   max_intensity_contribution=', maxIntensityContribution
    end if
    write(unit, '(A,E13.6,A,F10.7,A,F7.3)') '!  Solar_Flux=', solarFlux, '   Solar_Mu=', solarMu, '   Solar_Phi=', solarAzimuth
    write(unit, '(A,F7.4)')             '!  Lambertian_Surface_Albedo=', surfaceAlbedo
    write(unit, '(A)')                  '!  Output_Type= ' // outputType
  end subroutine writeHeader

  subroutine writeFluxFile
    integer :: i, j
    open(unit = 12, file = trim(outputFluxFile), status = "unknown")
    call writeHeader(12, "Flux", "Pixel Flux")
    write(12, '(A,F7.3,A,F7.3)') '!  Upwelling_Level=', zEdges(nz + 1), '   Downwelling_level=', zEdges(1)
    write(12, '(A)') '!   X      Y           Flux_Up             Flux_Down            Flux_Absorbed '
    write(12, '(A)') '!                  Mean     StdErr       Mean     StdErr       Mean     StdErr'
    write(12, '(A14,3(1X,2(1X,F9.4)))') '!  Average:   ', mMeans(1, :), mMeans(2, :), mMeans(3, :)
    do j = 1, ny
      do i = 1, nx
        write(12, '(2(F7.3),3(1X,2(1X,F9.4)))') sum(xEdges(i:i + 1)) / 2., sum(yEdges(j:j + 1)) / 2., &
                                                mUp(i, j, :), mDown(i, j, :), mAbs(i, j, :)
      end do
    end do
    close(12)
  end subroutine writeFluxFile

  subroutine writeProfileFile
    integer :: k
    open(unit = 12, file = trim(outputAbsProfFile), status = "unknown")
    call writeHeader(12, "Absorption Profile", "Absorption Profile")
    write(12, '(A)') '!   Z    Absorbed_Flux (flux/km) '
    write(12, '(A)') '!          Mean     StdErr '
    do k = 1, nz
      write(12, '(F7.3,1X,2(1X,F9.4))') 0.5 * (zEdges(k) + zEdges(k + 1)), mProfile(k, :)
    end do
    close(12)
  end subroutine writeProfileFile

  subroutine writeVolumeFile
    integer :: i, j, k
    open(unit = 12, file = trim(outputAbsVolumeFile), status = "unknown")
    call writeHeader(12, "3D Absorption Field", "Volume Absorption ")
    write(12, '(A)') '!    X       Y        Z       Absorbed_Flux (flux/km)'
    write(12, '(A)') '!                               Mean     StdErr '
    do i = 1, nx              ! x outermost, as downstream readers of the reference's file expect
      do j = 1, ny
        do k = 1, nz
          write(12, '(3(F7.3,1X),2(1X,F9.4))') sum(xEdges(i:i + 1)) / 2., sum(yEdges(j:j + 1)) / 2., &
                                               sum(zEdges(k:k + 1)) / 2., mVolume(i, j, k, :)
        end do
      end do
    end do
    close(12)
  end subroutine writeVolumeFile

  subroutine writeRadianceFile
    integer :: i, j, d
    open(unit = 12, file = trim(outputRadFile), status = "unknown")
    call writeHeader(12, "Radiance", "Pixel Radiance")
    write(12, '(A,F7.3,3(A,I4))') '!  RADIANCE AT Z=', zEdges(nz + 1), '   NXO=', nx, '   NYO=', ny, '   NDIR=', nDir
    write(12, '(A)') '!   X      Y         Radiance (Mean, StdErr)'
    do d = 1, nDir
      write(12, '(A,1X,F8.5,1X,F6.2,2X,A)') '! ', intensityMus(d), intensityPhis(d), '<- (mu,phi)'
      do j = 1, ny
        do i = 1, nx
          write(12, '(2(F7.3),2(1X,F9.4))') sum(xEdges(i:i + 1)) / 2., sum(yEdges(j:j + 1)) / 2., mRad(i, j, d, :)
        end do
      end do
    end do
    close(12)
  end subroutine writeRadianceFile

  ! Result file in netCDF classic format: same global attributes, dimensions (x, y, [z], [direction]) and variable
  ! names as the reference driver's writeResults_netcdf (:609-854), so that scripts reading its files read these.
  subroutine writeNetcdfFile
    use netcdf
    integer :: xDim, yDim, zDim, dirDim
    logical :: wantZ

    wantZ = reportAbsorptionProfile .or. reportVolumeAbsorption
    rc = nf90_create(trim(outputNetcdfFile), nf90_clobber, nc)
    if(rc /= nf90_NoErr) then
      print *, "Cannot create " // trim(outputNetcdfFile)
      return
    end if
    rc = nf90_put_att(nc, NF90_Global, "description", "Output from I3RC Community Monte Carlo Model")
    rc = nf90_put_att(nc, NF90_Global, "Domain_filename", trim(domainFileName))
    rc = nf90_put_att(nc, NF90_Global, "Surface_albedo", surfaceAlbedo)
    rc = nf90_put_att(nc, NF90_Global, "Total_number_of_photons", numPhotonsPerBatch * numBatches)
    rc = nf90_put_att(nc, NF90_Global, "Number_of_batches", numBatches)
    rc = nf90_put_att(nc, NF90_Global, "Solar_flux", solarFlux)
    rc = nf90_put_att(nc, NF90_Global, "Solar_mu", solarMu)
    rc = nf90_put_att(nc, NF90_Global, "Solar_phi", solarAzimuth)
    rc = nf90_put_att(nc, NF90_Global, "Random_number_seed", iseed)
    rc = nf90_put_att(nc, NF90_Global, "Phase_function_table_sizes", nPhaseIntervals)
    rc = nf90_put_att(nc, NF90_Global, "Algorithm", trim(merge("Ray_tracing      ", "Max_cross_section", useRayTracing)))
    call flagAndValue("Intensity_uses_hyrbid_phase_functions", useHybridPhaseFunsForIntenCalcs, &
                      "Hybrid_phase_function_width", hybridPhaseFunWidth)
    call flagAndValue("Intensity_uses_Russian_roulette", useRussianRouletteForIntensity, &
                      "Intensity_Russian_roulette_zeta_min", zetaMin)
    call flagAndValue("limited_intensity_contributions", limitIntensityContributions, &
                      "max_intensity_contribution", maxIntensityContribution)
    rc = nf90_put_att(nc, NF90_Global, "Cpu_time_total", t2)
    rc = nf90_put_att(nc, NF90_Global, "Cpu_time_setup", cpuSetup)
    rc = nf90_put_att(nc, NF90_Global, "Number_of_processors_used", numProcs)

    rc = nf90_def_dim(nc, "x", nx, xDim)
    rc = nf90_def_dim(nc, "y", ny, yDim)
    if(wantZ) rc = nf90_def_dim(nc, "z", nz, zDim)
    rc = nf90_def_var(nc, "x", nf90_float, xDim, v)
    rc = nf90_def_var(nc, "y", nf90_float, yDim, v)
    if(wantZ) rc = nf90_def_var(nc, "z", nf90_float, zDim, v)
    call definePair("fluxUp", (/ xDim, yDim /))
    call definePair("fluxDown", (/ xDim, yDim /))
    call definePair("fluxAbsorbed", (/ xDim, yDim /))
    if(reportAbsorptionProfile) call definePair("absorptionProfile", (/ zDim /))
    if(reportVolumeAbsorption)  call definePair("absorbedVolume", (/ xDim, yDim, zDim /))
    if(wantRadiance) then
      rc = nf90_def_dim(nc, "direction", nDir, dirDim)
      rc = nf90_def_var(nc, "intensityMus",  nf90_float, dirDim, v)
      rc = nf90_def_var(nc, "intensityPhis", nf90_float, dirDim, v)
      call definePair("intensity", (/ xDim, yDim, dirDim /))
    end if
    rc = nf90_enddef(nc)
    if(rc /= nf90_NoErr) print *, "netCDF output: definitions failed ", rc

    call put1("x", (xEdges(:nx) + xEdges(2:)) / 2)
    call put1("y", (yEdges(:ny) + yEdges(2:)) / 2)
    if(wantZ) call put1("z", (zEdges(:nz) + zEdges(2:)) / 2)
    call put2("fluxUp", mUp(:, :, 1));        call put2("fluxUp_StdErr", mUp(:, :, 2))
    call put2("fluxDown", mDown(:, :, 1));    call put2("fluxDown_StdErr", mDown(:, :, 2))
    call put2("fluxAbsorbed", mAbs(:, :, 1)); call put2("fluxAbsorbed_StdErr", mAbs(:, :, 2))
    if(reportAbsorptionProfile) then
      call put1("absorptionProfile", mProfile(:, 1)); call put1("absorptionProfile_StdErr", mProfile(:, 2))
    end if
    if(reportVolumeAbsorption) then
      call put3("absorbedVolume", mVolume(:, :, :, 1)); call put3("absorbedVolume_StdErr", mVolume(:, :, :, 2))
    end if
    if(wantRadiance) then
      call put1("intensityMus", intensityMus(:nDir)); call put1("intensityPhis", intensityPhis(:nDir))
      call put3("intensity", mRad(:, :, :, 1));       call put3("intensity_StdErr", mRad(:, :, :, 2))
    end if
    rc = nf90_close(nc)
    if(rc /= nf90_NoErr) print *, "netCDF output: close failed ", rc
  end subroutine writeNetcdfFile

    subroutine flagAndValue(flagName, flag, valueName, value)
      use netcdf
      character(len = *), intent(in) :: flagName, valueName
      logical,            intent(in) :: flag
      real,               intent(in) :: value
      rc = nf90_put_att(nc, NF90_Global, flagName, merge(1, 0, flag))
      rc = nf90_put_att(nc, NF90_Global, valueName, merge(value, 0., flag))
    end subroutine flagAndValue
    subroutine definePair(name, dims)
      use netcdf
      character(len = *), intent(in) :: name
      integer,            intent(in) :: dims(:)
      rc = nf90_def_var(nc, name, nf90_float, dims, v)
      rc = nf90_def_var(nc, name // "_StdErr", nf90_float, dims, v)
    end subroutine definePair
    subroutine put1(name, values)
      use netcdf
      character(len = *), intent(in) :: name
      real,               intent(in) :: values(:)
      rc = nf90_inq_varid(nc, name, v)
      if(rc == nf90_NoErr) rc = nf90_put_var(nc, v, values)
      if(rc /= nf90_NoErr) print *, "netCDF output: " // name, rc
    end subroutine put1
    subroutine put2(name, values)
      use netcdf
      character(len = *), intent(in) :: name
      real,               intent(in) :: values(:, :)
      rc = nf90_inq_varid(nc, name, v)
      if(rc == nf90_NoErr) rc = nf90_put_var(nc, v, values)
      if(rc /= nf90_NoErr) print *, "netCDF output: " // name, rc
    end subroutine put2
    subroutine put3(name, values)
      use netcdf
      character(len = *), intent(in) :: name
      real,               intent(in) :: values(:, :, :)
      rc = nf90_inq_varid(nc, name, v)
      if(rc == nf90_NoErr) rc = nf90_put_var(nc, v, values)
      if(rc /= nf90_NoErr) print *, "netCDF output: " // name, rc
    end subroutine put3
end program i3rcDriver
