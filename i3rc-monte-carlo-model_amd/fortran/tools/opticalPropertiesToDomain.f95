! Converts an SHDOM-like ASCII file of gridded optical properties to a domain file (netCDF classic) for read_Domain:
! the way to bring fields from other models into the Monte Carlo code without a netCDF toolchain.  Same interface as
! the reference's Tools/OpticalPropertiesToDomain.f95 (format: Tools/OpticalPropertiesToDomain.readme:27-65):
!   opticalPropertiesToDomain namelistFile          with   &fileNames PropFileName = "...", outputFileName = "..." /
! Input file:
!   T                                   first line begins with T ("tabulated phase functions")
!   Nx Ny Nz
!   delX delY Z(1) ... Z(Nz+1)          cell edges in z: one more level than SHDOM's grid-point files
!   numPhase
!   NumL chi_1 ... chi_NumL             one entry per phase function (may continue over lines); chi_l = (2l + 1) x the
!                                       moment this code uses, chi_0 = 1 is not listed
!   IX IY IZ Temp Extinct Albedo Iphase one line per cell, any order; cells not listed stay empty
program opticalPropertiesToDomain
  use ErrorMessages
  use UserInterface
  use scatteringPhaseFunctions
  use opticalProperties
  implicit none
  character(len = 256) :: PropFileName = "", outputFileName = "", namelistFile
  namelist /fileNames/ PropFileName, outputFileName
  character(len = 8) :: firstWord
  integer :: nx, ny, nz, numPhase, numL, i, k, ix, iy, iz, iPhase, ioStatus, nCells
  real    :: delX, delY, temperature, ext, ssa
  real,    allocatable :: zLevels(:), chi(:), extinction(:, :, :), albedo(:, :, :)
  integer, allocatable :: phaseIndex(:, :, :)
  type(phaseFunction), allocatable :: phaseFunctions(:)
  type(ErrorMessage)       :: status
  type(phaseFunctionTable) :: table
  type(domain)             :: field

  namelistFile = getOneArgument()
  open(unit = 11, file = trim(namelistFile), status = "old", action = "read")
  read(11, nml = fileNames)
  close(11)
  if(len_trim(PropFileName) == 0 .or. len_trim(outputFileName) == 0) error stop "fileNames: PropFileName and outputFileName are needed"

  open(unit = 12, file = trim(PropFileName), status = "old", action = "read")
  read(12, *) firstWord
  if(firstWord(1:1) /= "T") error stop "property file: the first line must begin with T (tabulated phase functions)"
  read(12, *) nx, ny, nz
  if(nx < 1 .or. ny < 1 .or. nz < 1) error stop "property file: bad grid size"
  allocate(zLevels(nz + 1))
  read(12, *) delX, delY, zLevels
  read(12, *) numPhase
  if(numPhase < 1) error stop "property file: at least one phase function is needed"
  allocate(phaseFunctions(numPhase))
  do k = 1, numPhase
    read(12, *) numL                                        ! list-directed reads continue over line ends, so read the
    backspace(12)                                           ! count first and then the whole entry in one statement
    allocate(chi(numL))
    read(12, *) numL, chi
    phaseFunctions(k) = new_PhaseFunction(chi / (/ (real(2 * i + 1), i = 1, numL) /), status = status)
    call printStatus(status)
    deallocate(chi)
  end do
  table = new_PhaseFunctionTable(phaseFunctions, key = (/ (real(k), k = 1, numPhase) /), &
                                 tableDescription = "Phase functions of " // trim(PropFileName), status = status)
  call printStatus(status)

  allocate(extinction(nx, ny, nz), albedo(nx, ny, nz), phaseIndex(nx, ny, nz))
  extinction = 0.; albedo = 0.; phaseIndex = 1
  nCells = 0
  do
    read(12, *, iostat = ioStatus) ix, iy, iz, temperature, ext, ssa, iPhase
    if(ioStatus /= 0) exit
    if(ix < 1 .or. ix > nx .or. iy < 1 .or. iy > ny .or. iz < 1 .or. iz > nz) error stop "property file: cell index outside the grid"
    if(iPhase < 1 .or. iPhase > numPhase) error stop "property file: phase function index outside the table"
    extinction(ix, iy, iz) = ext; albedo(ix, iy, iz) = ssa; phaseIndex(ix, iy, iz) = iPhase
    nCells = nCells + 1
  end do
  close(12)

  field = new_Domain(xPosition = delX * (/ (real(i), i = 0, nx) /), yPosition = delY * (/ (real(i), i = 0, ny) /), &
                     zPosition = zLevels, status = status)
  call printStatus(status)
  call addOpticalComponent(field, "Optical properties from " // trim(PropFileName), extinction, albedo, phaseIndex, table, &
                           status = status)
  call printStatus(status)
  call write_Domain(field, trim(outputFileName), status = status)
  call printStatus(status)
  print '(A, A, A, I0, A, I0, A)', "wrote ", trim(outputFileName), ": ", nCells, " cells listed, ", numPhase, " phase functions"
  call finalize_Domain(field)
end program opticalPropertiesToDomain
