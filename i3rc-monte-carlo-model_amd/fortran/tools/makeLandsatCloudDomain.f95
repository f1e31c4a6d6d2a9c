! Writes the I3RC phase-1 Landsat-scene domain (128 x 128 columns of 30 m; per column an optical depth and a
! geometrical thickness from the scene-43 files of the I3RC case definition) as a netCDF classic file for read_Domain.
! Recipe: I3RC-Examples/i3rcLandsatCloud.f95:27-35 (geometry: 20 m layers from z = 200 m, as many as the thickest
! column needs), :70-83 (files: 128 rows of 128f7.2, thickness in km), :92-104 (a column's cloud fills its lowest
! nint(thickness / deltaZ) layers with extinction tau / (layers deltaZ); clear cells keep extinction 0, albedo 0 and
! phase function index 0), restated through the shell's own API.
!   makeLandsatCloudDomain dataDirectory outputFile [singleScatteringAlbedo] [nLayers]
! nLayers (default 119 = the reference's) re-bins the same 2380 m into thicker layers: 36 gives BASELINE.json's
! labelled 128 x 128 x 36 synthetic (cloud in the lowest max(1, nint(thickness / deltaZ)) layers of a cloudy column).
program makeLandsatCloudDomain
  use ErrorMessages
  use UserInterface
  use scatteringPhaseFunctions
  use opticalProperties
  implicit none
  integer, parameter :: nx = 128, ny = 128, nMoments = 299, referenceLayers = 119
  real,    parameter :: deltaXY = 30., g = 0.85, cloudBase = 200., maxThickness = 2380.
  character(len = 256) :: dataDir, fileName, argument
  integer :: nLayers, i, j, n
  real    :: ssa, deltaZ, opticalDepth(nx, ny), thickness(nx, ny)
  real,    allocatable :: extinction(:, :, :), albedo(:, :, :)
  integer, allocatable :: phaseIndex(:, :, :)
  type(ErrorMessage)       :: status
  type(phaseFunction)      :: hg
  type(phaseFunctionTable) :: table
  type(domain)             :: cloud

  if(command_argument_count() < 2) error stop "usage: makeLandsatCloudDomain dataDirectory outputFile [ssa] [nLayers]"
  call get_command_argument(1, dataDir); call get_command_argument(2, fileName)
  ssa = 1.; nLayers = referenceLayers
  if(command_argument_count() >= 3) then
    call get_command_argument(3, argument); read(argument, *) ssa
  end if
  if(command_argument_count() >= 4) then
    call get_command_argument(4, argument); read(argument, *) nLayers
  end if
  if(nLayers < 1) error stop "nLayers must be positive"
  deltaZ = 20.
  if(nLayers /= referenceLayers) deltaZ = maxThickness / real(nLayers)

  open(unit = 10, file = trim(dataDir) // "/scene43.tau.128x128", status = "old", action = "read")
  do j = 1, ny
    read(10, '(128f7.2)') opticalDepth(:, j)
  end do
  close(10)
  open(unit = 10, file = trim(dataDir) // "/scene43.dz.128x128", status = "old", action = "read")
  do j = 1, ny
    read(10, '(128f7.2)') thickness(:, j)
  end do
  close(10)
  thickness = thickness * 1000.                            ! km -> m
  if(any((thickness > 0.) .neqv. (opticalDepth > 0.))) print *, "warning: thickness and optical depth disagree on cloudy columns"

  allocate(extinction(nx, ny, nLayers), albedo(nx, ny, nLayers), phaseIndex(nx, ny, nLayers))
  extinction = 0.; albedo = 0.; phaseIndex = 0
  do j = 1, ny
    do i = 1, nx
      if(opticalDepth(i, j) > tiny(1.)) then
        n = nint(thickness(i, j) / deltaZ)
        if(nLayers /= referenceLayers) n = max(1, n)
        n = min(n, nLayers)
        if(n > 0) extinction(i, j, :n) = opticalDepth(i, j) / (n * deltaZ)
      end if
    end do
  end do
  where(extinction > 0.)
    albedo = ssa
    phaseIndex = 1
  end where

  hg = new_PhaseFunction(g**(/ (i, i = 1, nMoments) /), status = status)
  call printStatus(status)
  table = new_PhaseFunctionTable((/ hg /), key = (/ 1. /), tableDescription = "Henyey-Greenstein with g = 0.85", status = status)
  call printStatus(status)
  cloud = new_Domain(xPosition = deltaXY * (/ (real(i), i = 0, nx) /), yPosition = deltaXY * (/ (real(i), i = 0, ny) /), &
                     zPosition = deltaZ * (/ (real(i), i = 0, nLayers) /) + cloudBase, status = status)
  call printStatus(status)
  call addOpticalComponent(cloud, "cloud", extinction, albedo, phaseIndex, table, status = status)
  call printStatus(status)
  call write_Domain(cloud, trim(fileName), status = status)
  call printStatus(status)
  print '(A, A, A, F8.4, A, F6.3)', "wrote ", trim(fileName), ": mean column optical depth ", &
        real(sum(real(extinction, kind(1.d0))) * deltaZ / (nx * ny)), ", cloudy fraction of cells ", real(count(extinction > 0.)) / size(extinction)
  call finalize_Domain(cloud)
end program makeLandsatCloudDomain
