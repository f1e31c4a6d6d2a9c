! Writes the I3RC phase-1 radar-cloud domain (640 x 1 x 54 cells of 50 m x 32 km x 45 m; the optical depth of every
! cell comes from the MMCR retrieval file of the I3RC case definition) as a netCDF classic file for read_Domain.
! Recipe: I3RC-Examples/i3rcRadarCloud.f95:28-31 (geometry), :107-125 (file row j fills layer nLayers + 1 - j,
! optical depth / deltaZ), :60-105 (phase functions: Henyey-Greenstein g = 0.85 with 299 moments, Deirmendjian C1
! tabulated at 1801 angles, or C1 from its Legendre expansion whose file lists (2l + 1) chi_l), restated through
! the shell's own API.
!   makeRadarCloudDomain dataDirectory outputFile [singleScatteringAlbedo] [hg | c1 | c1legendre]
! dataDirectory holds mmcr_tau_32km_020898 (54 rows of 640f8.3), C.1_PF (angle in degrees, value) and C.1_leg_coef.
program makeRadarCloudDomain
  use ErrorMessages
  use UserInterface
  use scatteringPhaseFunctions
  use opticalProperties
  implicit none
  integer, parameter :: nColumns = 640, nLayers = 54, nMoments = 299, nAngles = 1801
  real,    parameter :: deltaX = 50., deltaZ = 45., g = 0.85
  character(len = 256) :: dataDir, fileName, argument
  character(len = 16)  :: phaseKind
  real    :: ssa, angle(nAngles), value(nAngles), moments(0:nMoments)
  real    :: extinction(nColumns, 1, nLayers), albedo(nColumns, 1, nLayers)
  integer :: phaseIndex(nColumns, 1, nLayers), i, layer
  type(ErrorMessage)       :: status
  type(phaseFunction)      :: pf
  type(phaseFunctionTable) :: table
  type(domain)             :: cloud

  if(command_argument_count() < 2) error stop "usage: makeRadarCloudDomain dataDirectory outputFile [ssa] [hg|c1|c1legendre]"
  call get_command_argument(1, dataDir); call get_command_argument(2, fileName)
  ssa = 1.; phaseKind = "hg"
  if(command_argument_count() >= 3) then
    call get_command_argument(3, argument); read(argument, *) ssa
  end if
  if(command_argument_count() >= 4) call get_command_argument(4, phaseKind)

  select case(trim(phaseKind))
    case("hg")
      pf = new_PhaseFunction(g**(/ (i, i = 1, nMoments) /), status = status)
      call printStatus(status)
      table = new_PhaseFunctionTable((/ pf /), key = (/ 1. /), tableDescription = "Henyey-Greenstein with g = 0.85", &
                                     status = status)
    case("c1")
      open(unit = 10, file = trim(dataDir) // "/C.1_PF", status = "old", action = "read")
      do i = 1, nAngles
        read(10, *) angle(i), value(i)
      end do
      close(10)
      table = new_PhaseFunctionTable(angle * acos(-1.) / 180., spread(value, 2, nCopies = 1), key = (/ 1. /), &
                                     tableDescription = "Deirmendjian C1", status = status)
    case("c1legendre")
      open(unit = 10, file = trim(dataDir) // "/C.1_leg_coef", status = "old", action = "read")
      do i = 0, nMoments                                   ! the file starts with chi_0 = 1
        read(10, *) moments(i)
      end do
      close(10)
      pf = new_PhaseFunction(moments(1:) / (/ (real(2 * i + 1), i = 1, nMoments) /), status = status)
      call printStatus(status)
      table = new_PhaseFunctionTable((/ pf /), key = (/ 1. /), tableDescription = "Deirmendjian C1", status = status)
    case default
      error stop "phase function must be hg, c1 or c1legendre"
  end select
  call printStatus(status)

  open(unit = 10, file = trim(dataDir) // "/mmcr_tau_32km_020898", status = "old", action = "read")
  do layer = nLayers, 1, -1                                ! the file is written from the top of the cloud down
    read(10, '(640f8.3)') extinction(:, 1, layer)
  end do
  close(10)
  extinction = extinction / deltaZ                         ! optical depth per cell -> extinction
  albedo = ssa; phaseIndex = 1
  cloud = new_Domain(xPosition = deltaX * (/ (real(i), i = 0, nColumns) /), yPosition = (/ 0., deltaX * nColumns /), &
                     zPosition = deltaZ * (/ (real(i), i = 0, nLayers) /), status = status)
  call printStatus(status)
  call addOpticalComponent(cloud, "cloud", extinction, albedo, phaseIndex, table, status = status)
  call printStatus(status)
  call write_Domain(cloud, trim(fileName), status = status)
  call printStatus(status)
  print '(A, A, A, F8.4, A, I0, A)', "wrote ", trim(fileName), ": mean column optical depth ", &
        real(sum(real(extinction, kind(1.d0))) * deltaZ / nColumns), ", ", count(extinction > 0.), " cloudy cells"
  call finalize_Domain(cloud)
end program makeRadarCloudDomain
