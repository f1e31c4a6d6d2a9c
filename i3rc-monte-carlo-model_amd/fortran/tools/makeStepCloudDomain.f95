! Writes the I3RC phase-1 step-cloud domain (32 columns, optical depth 2 | 18, HG g = 0.85) as a netCDF classic
! file that read_Domain / the reference's monteCarloDriver can load.  The recipe is the one of
! I3RC-Examples/i3rcStepCloud.f95:27-75, re-stated here through the shell's own API.
!   makeStepCloudDomain fileName [nLayers] [singleScatteringAlbedo]
program makeStepCloudDomain
  use ErrorMessages
  use UserInterface
  use scatteringPhaseFunctions
  use opticalProperties
  implicit none
  integer, parameter :: nColumns = 32, nMoments = 64
  real,    parameter :: domainSize = 500., thickness = 250., g = 0.85
  integer :: nLayers, i
  real    :: ssa, deltaX, deltaZ
  character(len = 256) :: fileName, argument
  real,    dimension(:, :, :), allocatable :: extinction, albedo
  integer, dimension(:, :, :), allocatable :: phaseIndex
  type(ErrorMessage)       :: status
  type(phaseFunction)      :: hg
  type(phaseFunctionTable) :: table
  type(domain)             :: cloud

  if(command_argument_count() < 1) stop "usage: makeStepCloudDomain fileName [nLayers] [ssa]"
  call get_command_argument(1, fileName)
  nLayers = 32; ssa = 1.
  if(command_argument_count() >= 2) then
    call get_command_argument(2, argument); read(argument, *) nLayers
  end if
  if(command_argument_count() >= 3) then
    call get_command_argument(3, argument); read(argument, *) ssa
  end if
  allocate(extinction(nColumns, 1, nLayers), albedo(nColumns, 1, nLayers), phaseIndex(nColumns, 1, nLayers))
  hg = new_PhaseFunction(g**(/ (i, i = 1, nMoments) /), status = status)
  table = new_PhaseFunctionTable((/ hg /), key = (/ 1. /), tableDescription = "Henyey-Greenstein with g = 0.85", status = status)
  call printStatus(status)
  deltaX = domainSize / real(nColumns); deltaZ = thickness / real(nLayers)
  extinction(:, 1, :) = spread((/ (2, i = 1, nColumns / 2), (18, i = 1, nColumns / 2) /), dim = 2, nCopies = nLayers) / thickness
  albedo = ssa; phaseIndex = 1
  cloud = new_Domain(xPosition = deltaX * (/ 0., (real(i), i = 1, nColumns) /), yPosition = (/ 0., 500.0 /), &
                     zPosition = deltaZ * (/ 0., (real(i), i = 1, nLayers) /), status = status)
  call addOpticalComponent(cloud, "cloud", extinction, albedo, phaseIndex, table, status = status)
  call printStatus(status)
  call write_Domain(cloud, trim(fileName), status = status)
  call printStatus(status)
  print *, "wrote ", trim(fileName)
  call finalize_Domain(cloud)
end program makeStepCloudDomain
