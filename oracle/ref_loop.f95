! oracle/ref_loop.f95 -- test infrastructure; THIS repository's own caller (no line of it is the reference's).
!
! It drives the REFERENCE'S OWN photon loop: oracle/Makefile, target _ref_loop, compiles every module of /root/reference/Code and
! Integrators/monteCarloRadiativeTransfer.f95 UNMODIFIED and in place -- no file of them is copied, edited or stubbed -- and links
! them with this program.  The one thing in that build that is not the reference's is `module netcdf`: netCDF-Fortran is not in this
! image, and three of the reference's modules say `use netcdf` for their file routines; the module they find is this repository's
! own netCDF-classic implementation (i3rc-monte-carlo-model_amd/fortran/netcdf_classic.f95, written from the format specification:
! SURVEY.md section 8f row 1 -- part of the product, the module the reference's unchanged drivers are linked against).  This
! program builds its domains in memory and never calls it.  By the letter of the build rules a reference that needs a library the
! image lacks counts as unbuildable, so this is NOT offered as oracle/_ref in the rules' sense and lifts no "unpinned" label by
! itself; it is what it is: the reference's loop, every line of it, run here, for oracle/integrator.c to be held against
! (tests/golden/make_ref_loop.py -> tests/golden/ref_loop.npz, tests/test_ref_loop.py).
!
! usage: ref_loop <case file> <result file> [<domain file to write> [<domain file to read back>]]      case and result: little-endian streams of 4-byte words:
!   case:   nx ny nz ncomp | xEdges yEdges zEdges | per component: nEntries, per entry (nCoef, coef(1:nCoef) -- or -n, angles(n), values(n)), ext(nx,ny,nz),
!           ssa(nx,ny,nz), phaseFunctionIndex(nx,ny,nz) | surfaceAlbedo useRayTracing useRussianRoulette
!           useRussianRouletteForIntensity zetaMin useHybrid hybridWidth numOrdersOrig limitContributions maxContribution
!           nInverse nForward nDir mus(nDir) phis(nDir) | nxs nys [xs(nxs+1) ys(nys+1) reflectance(nxs,nys)] |
!           solarMu solarAzimuth nBatches nPhotons seed1 seed2OfFirstBatch dumpTables
!   result: per batch fluxUp(nx,ny) fluxDown fluxAbsorbed absorbedProfile(nz) volumeAbsorption(nx,ny,nz) [intensity(nx,ny,nDir)];
!           then, if dumpTables: per component the inverse table (nInverse, nEntries) and the forward table (nForward, nEntries)
program refLoop
  use ErrorMessages
  use RandomNumbers
  use scatteringPhaseFunctions
  use inversePhaseFunctions
  use opticalProperties
  use surfaceProperties
  use monteCarloIllumination
  use monteCarloRadiativeTransfer
  implicit none

  character(len = 512) :: caseFile, resultFile, domainFile
  integer :: nx, ny, nz, ncomp, c, e, nEntries, nCoef, b, k
  integer :: useRT, useRR, useRRI, useHybrid, numOrdersOrig, limitC, nInv, nFwd, nDir, nxs, nys
  integer :: nBatches, nPhotons, seed1, seed2, dumpTables
  real    :: albedo, zetaMin, hybridWidth, maxContribution, solarMu, solarAzimuth
  real, allocatable :: xE(:), yE(:), zE(:), coef(:), tabValue(:), ext(:, :, :), ssa(:, :, :), mus(:), phis(:)
  real, allocatable :: xs(:), ys(:), refl(:, :), brdf(:, :, :)
  integer, allocatable :: pf(:, :, :)
  real, allocatable :: fluxUp(:, :), fluxDown(:, :), fluxAbsorbed(:, :), profile(:), volume(:, :, :), intensity(:, :, :)
  real, allocatable :: inverse(:, :), forward(:, :), angles(:)
  real, parameter   :: Pi = 3.14159265358979312   ! (the integrator's own constant, for the forward table's angles)
  type(ErrorMessage)          :: status
  type(phaseFunction), allocatable :: functions(:)
  type(phaseFunctionTable), allocatable :: tables(:), domainTables(:)
  real, allocatable    :: totalExt(:, :, :), cumExt(:, :, :, :), ssa4(:, :, :, :)
  integer, allocatable :: pf4(:, :, :, :)
  type(domain)                :: theDomain
  type(surfaceDescription)    :: surface
  type(integrator)            :: mcIntegrator
  type(randomNumberSequence)  :: randoms
  type(photonStream)          :: photons

  call get_command_argument(1, caseFile)
  call get_command_argument(2, resultFile)
  open(unit = 10, file = trim(caseFile), access = "stream", form = "unformatted", status = "old", action = "read")
  open(unit = 11, file = trim(resultFile), access = "stream", form = "unformatted", status = "replace", action = "write")
  call initializeState(status)

  read(10) nx, ny, nz, ncomp
  allocate(xE(nx + 1), yE(ny + 1), zE(nz + 1), ext(nx, ny, nz), ssa(nx, ny, nz), pf(nx, ny, nz), tables(ncomp))
  read(10) xE, yE, zE
  theDomain = new_Domain(xE, yE, zE, status)
  call check("new_Domain")
  do c = 1, ncomp
    read(10) nEntries
    allocate(functions(nEntries))
    do e = 1, nEntries
      read(10) nCoef
      if(nCoef >= 0) then         ! Legendre coefficients P1 ... PnCoef
        allocate(coef(nCoef))
        read(10) coef
        functions(e) = new_PhaseFunction(coef, status = status)
        deallocate(coef)
      else                        ! -nCoef scattering angles (radians), then as many values
        allocate(coef(-nCoef), tabValue(-nCoef))
        read(10) coef, tabValue
        functions(e) = new_PhaseFunction(coef, tabValue, status = status)
        deallocate(coef, tabValue)
      end if
      call check("new_PhaseFunction")
    end do
    tables(c) = new_PhaseFunctionTable(functions, key = (/ (real(e), e = 1, nEntries) /), status = status)
    call check("new_PhaseFunctionTable")
    deallocate(functions)
    read(10) ext, ssa, pf
    call addOpticalComponent(theDomain, "component", ext, ssa, pf, tables(c), status = status)
    call check("addOpticalComponent")
  end do

  ! a third argument: the domain as a file (write_Domain, Code/opticalProperties.f95:554-706) -- and a fourth: that file read back
  ! (read_Domain, :708-871) into the domain the rest of the program goes on with
  if(command_argument_count() >= 3) then
    call get_command_argument(3, domainFile)
    call write_Domain(theDomain, trim(domainFile), status)
    call check("write_Domain")
  end if
  if(command_argument_count() >= 4) then
    call get_command_argument(4, domainFile)
    call finalize_Domain(theDomain)
    call read_Domain(trim(domainFile), theDomain, status)
    call check("read_Domain")
  end if

  read(10) albedo, useRT, useRR, useRRI, zetaMin, useHybrid, hybridWidth, numOrdersOrig, limitC, maxContribution, nInv, nFwd, nDir
  allocate(mus(nDir), phis(nDir))
  read(10) mus, phis
  read(10) nxs, nys
  if(nxs > 0) allocate(xs(nxs + 1), ys(nys + 1), refl(nxs, nys), brdf(1, nxs, nys))
  if(nxs > 0) read(10) xs, ys, refl
  read(10) solarMu, solarAzimuth, nBatches, nPhotons, seed1, seed2, dumpTables
  if(nBatches > 0) then         ! (no batches: the tables only)
  mcIntegrator = new_Integrator(theDomain, status = status)
  call check("new_Integrator")
  call specifyParameters(mcIntegrator, minInverseTableSize = nInv, minForwardTableSize = nFwd, surfaceAlbedo = albedo,        &
                         useRayTracing = useRT /= 0, useRussianRoulette = useRR /= 0, status = status)
  call check("specifyParameters")
  if(nxs > 0) then
    brdf(1, :, :) = refl
    surface = new_SurfaceDescription(brdf, xs, ys, status)
    call check("new_SurfaceDescription")
    call specifyParameters(mcIntegrator, surfaceBDRF = surface, status = status)
    call check("specifyParameters (surface)")
  end if
  if(nDir > 0) then
    call specifyParameters(mcIntegrator, intensityMus = mus, intensityPhis = phis, computeIntensity = .true.,                &
                           useRussianRouletteForIntensity = useRRI /= 0, zetaMin = zetaMin,                                &
                           useHybridPhaseFunsForIntenCalcs = useHybrid /= 0, hybridPhaseFunWidth = hybridWidth,            &
                           numOrdersOrigPhaseFunIntenCalcs = numOrdersOrig,                                                &
                           limitIntensityContributions = limitC /= 0, maxIntensityContribution = maxContribution, status = status)
    call check("specifyParameters (intensity)")
  end if

  end if
  allocate(fluxUp(nx, ny), fluxDown(nx, ny), fluxAbsorbed(nx, ny), profile(nz), volume(nx, ny, nz), intensity(nx, ny, max(nDir, 1)))
  do b = 1, nBatches     ! the drivers' batch loop: Example-Drivers/monteCarloDriver.f95:283-326
    randoms = new_RandomNumberSequence(seed = (/ seed1, seed2 + b - 1 /))
    photons = new_PhotonStream(solarMu, solarAzimuth, numberOfPhotons = nPhotons, randomNumbers = randoms, status = status)
    call check("new_PhotonStream")
    call computeRadiativeTransfer(mcIntegrator, randoms, photons, status)
    call check("computeRadiativeTransfer")
    if(nDir > 0) then
      call reportResults(mcIntegrator, fluxUp = fluxUp, fluxDown = fluxDown, fluxAbsorbed = fluxAbsorbed, absorbedProfile = profile, &
                         volumeAbsorption = volume, intensity = intensity, status = status)
    else
      call reportResults(mcIntegrator, fluxUp = fluxUp, fluxDown = fluxDown, fluxAbsorbed = fluxAbsorbed, absorbedProfile = profile, &
                         volumeAbsorption = volume, status = status)
    end if
    call check("reportResults")
    write(11) fluxUp, fluxDown, fluxAbsorbed, profile, volume
    if(nDir > 0) write(11) intensity
    call finalize_PhotonStream(photons)
    call finalize_RandomNumberSequence(randoms)
  end do

  if(dumpTables /= 0) then   ! what tabulateInversePhaseFunctions (:1809-1861) and tabulateForwardPhaseFunctions (:1863-1923) call,
    ! on the tables as new_Integrator gets them (:221-223: every copy of a TABULATED phase function normalises it again,
    ! Code/scatteringPhaseFunctions.f95:411-442, so the integrator's tables are not quite the ones handed to addOpticalComponent)
    allocate(totalExt(nx, ny, nz), cumExt(nx, ny, nz, ncomp), ssa4(nx, ny, nz, ncomp), pf4(nx, ny, nz, ncomp), domainTables(ncomp))
    call getOpticalPropertiesByComponent(theDomain, totalExt, cumExt, ssa4, pf4, domainTables, status)
    call check("getOpticalPropertiesByComponent")
    tables(:) = domainTables(:)
    allocate(angles(nFwd))
    angles = (/ (real(k), k = 0, nFwd - 1) /) / real(nFwd - 1) * Pi
    do c = 1, ncomp
      call getInfo_PhaseFunctionTable(tables(c), nEntries = nEntries, status = status)
      allocate(inverse(nInv, nEntries), forward(nFwd, nEntries))
      call computeInversePhaseFuncTable(tables(c), inverse, status)
      call check("computeInversePhaseFuncTable")
      call getPhaseFunctionValues(tables(c), angles, forward, status)
      call check("getPhaseFunctionValues")
      write(11) inverse, forward
      deallocate(inverse, forward)
    end do
  end if
  close(11)

contains
  subroutine check(what)
    character(len = *), intent(in) :: what
    if(stateIsFailure(status)) then
      print *, "ref_loop: " // what // " failed: " // trim(getCurrentMessage(status))
      error stop 1
    end if
  end subroutine check
end program refLoop
