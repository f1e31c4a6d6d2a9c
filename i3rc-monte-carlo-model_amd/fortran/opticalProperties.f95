! Fortran-95 shell of the MI355X photon-tracing integrator -- the 3-D domain of optical properties.
! Public interface of the reference's module opticalProperties (Code/opticalProperties.f95:84-90): domain,
! new_Domain, addOpticalComponent / replaceOpticalComponent / deleteOpticalComponent, getInfo_Domain,
! getOpticalPropertiesByComponent (the layout the GPU copies), read_Domain / write_Domain (netCDF classic
! files with the reference's variable names), finalize_Domain.
module opticalProperties
  use CharacterUtils,           only: IntToChar
  use ErrorMessages,            only: ErrorMessage, stateIsFailure, setStateToFailure, setStateToWarning, setStateToSuccess
  use scatteringPhaseFunctions, only: phaseFunctionTable, getInfo_PhaseFunctionTable, copy_PhaseFunctionTable, &
                                      finalize_PhaseFunctionTable, isReady_PhaseFunctionTable,                &
                                      add_PhaseFunctionTable, read_PhaseFunctionTable
  use netcdf
  implicit none
  private
  integer, parameter :: nameLength = 256

  type opticalComponent
    private
    character(len = nameLength)          :: name = ""
    integer                              :: zLevelBase = 0
    logical                              :: horizontallyUniform = .false.
    real,    dimension(:, :, :), pointer :: extinction             => null()
    real,    dimension(:, :, :), pointer :: singleScatteringAlbedo => null()
    integer, dimension(:, :, :), pointer :: phaseFunctionIndex     => null()
    type(phaseFunctionTable)             :: table
  end type opticalComponent

  type domain
    private
    real, dimension(:), pointer :: xPosition => null(), yPosition => null(), zPosition => null()
    logical                     :: xyRegularlySpaced = .false., zRegularlySpaced = .false.
    type(opticalComponent), dimension(:), pointer :: components => null()
  end type domain

  interface addOpticalComponent
    module procedure addComponent3D, addComponent1D
  end interface
  interface replaceOpticalComponent
    module procedure replaceComponent3D, replaceComponent1D
  end interface

  public :: domain
  public :: new_Domain, getInfo_Domain, write_Domain, read_Domain, finalize_Domain, &
            addOpticalComponent, deleteOpticalComponent, replaceOpticalComponent,   &
            getOpticalPropertiesByComponent
contains
  ! ------------------------------------------------------------------------------------------------
  pure logical function evenlySpaced(edges)
    real, dimension(:), intent(in) :: edges
    integer :: n
    n = size(edges)
    evenlySpaced = all(abs((edges(2:) - edges(:n - 1)) - (edges(2) - edges(1))) <= 2 * spacing(edges(2:)))
  end function evenlySpaced

  function new_Domain(xPosition, yPosition, zPosition, status)
    real, dimension(:), intent(in   ) :: xPosition, yPosition, zPosition
    type(ErrorMessage), intent(inout) :: status
    type(domain)                      :: new_Domain
    integer :: nx, ny, nz
    nx = size(xPosition); ny = size(yPosition); nz = size(zPosition)
    if(any(xPosition(2:) - xPosition(:nx - 1) <= 0.) .or. any(yPosition(2:) - yPosition(:ny - 1) <= 0.) .or. &
       any(zPosition(2:) - zPosition(:nz - 1) <= 0.)) then
      call setStateToFailure(status, "new_Domain: Positions must be increasing, unique.")
      return
    end if
    allocate(new_Domain%xPosition(nx), new_Domain%yPosition(ny), new_Domain%zPosition(nz))
    new_Domain%xPosition(:) = xPosition(:)
    new_Domain%yPosition(:) = yPosition(:)
    new_Domain%zPosition(:) = zPosition(:)
    new_Domain%xyRegularlySpaced = evenlySpaced(xPosition) .and. evenlySpaced(yPosition)
    new_Domain%zRegularlySpaced  = evenlySpaced(zPosition)
    call setStateToSuccess(status)
  end function new_Domain

  logical function isValid(thisDomain)
    type(domain), intent(in) :: thisDomain
    isValid = associated(thisDomain%xPosition) .and. associated(thisDomain%yPosition) .and. associated(thisDomain%zPosition)
  end function isValid

  ! ------------------------------------------------------------------------------------------------
  ! Components
  ! ------------------------------------------------------------------------------------------------
  subroutine validateComponent(thisDomain, extinction, singleScatteringAlbedo, phaseFunctionIndex, table, zLevelBase, status)
    type(domain),                intent(in   ) :: thisDomain
    real,    dimension(:, :, :), intent(in   ) :: extinction, singleScatteringAlbedo
    integer, dimension(:, :, :), intent(in   ) :: phaseFunctionIndex
    type(phaseFunctionTable),    intent(in   ) :: table
    integer,                     intent(in   ) :: zLevelBase
    type(ErrorMessage),          intent(inout) :: status
    integer :: nx, ny, nz, nEntries

    if(.not. isValid(thisDomain)) then
      call setStateToFailure(status, "validateOpticalComponent: domain hasn't been initialized.")
      return
    end if
    nx = size(thisDomain%xPosition) - 1; ny = size(thisDomain%yPosition) - 1; nz = size(thisDomain%zPosition) - 1
    if(any(shape(extinction) /= shape(singleScatteringAlbedo)) .or. any(shape(extinction) /= shape(phaseFunctionIndex))) &
      call setStateToFailure(status, "validateOpticalComponent: optical property grids must be the same size.")
    if(.not. ((size(extinction, 1) == nx .and. size(extinction, 2) == ny) .or. &
              (size(extinction, 1) == 1  .and. size(extinction, 2) == 1))) &
      call setStateToFailure(status, "validateOpticalComponent: optical property grids don't match the domain in x, y.")
    if(zLevelBase < 1 .or. zLevelBase + size(extinction, 3) - 1 > nz) &
      call setStateToFailure(status, "validateOpticalComponent: optical property grids don't fit the domain in z.")
    if(any(extinction < 0.)) call setStateToFailure(status, "validateOpticalComponent: extinction must be >= 0.")
    if(any(singleScatteringAlbedo < 0.) .or. any(singleScatteringAlbedo > 1.)) &
      call setStateToFailure(status, "validateOpticalComponent: singleScatteringAlbedo must be between 0 and 1")
    if(.not. isReady_PhaseFunctionTable(table)) then
      call setStateToFailure(status, "validateOpticalComponent: phase function table is not ready.")
    else
      call getInfo_PhaseFunctionTable(table, nEntries = nEntries, status = status)
      if(any(phaseFunctionIndex < 0) .or. any(phaseFunctionIndex > nEntries)) &
        call setStateToFailure(status, "validateOpticalComponent: phase function index is out of bounds")
      if(any(phaseFunctionIndex == 0 .and. extinction > 0.)) &
        call setStateToFailure(status, "validateOpticalComponent: phase function index is 0 where extinction is non-zero")
    end if
    if(.not. stateIsFailure(status)) call setStateToSuccess(status)
  end subroutine validateComponent

  subroutine fillComponent(c, name, extinction, singleScatteringAlbedo, phaseFunctionIndex, zLevelBase, table, uniform)
    type(opticalComponent),      intent(inout) :: c
    character(len = *),          intent(in   ) :: name
    real,    dimension(:, :, :), intent(in   ) :: extinction, singleScatteringAlbedo
    integer, dimension(:, :, :), intent(in   ) :: phaseFunctionIndex
    integer,                     intent(in   ) :: zLevelBase
    type(phaseFunctionTable),    intent(in   ) :: table
    logical,                     intent(in   ) :: uniform
    integer :: n1, n2, n3
    n1 = size(extinction, 1); n2 = size(extinction, 2); n3 = size(extinction, 3)
    allocate(c%extinction(n1, n2, n3), c%singleScatteringAlbedo(n1, n2, n3), c%phaseFunctionIndex(n1, n2, n3))
    c%extinction = extinction
    c%singleScatteringAlbedo = singleScatteringAlbedo
    c%phaseFunctionIndex = phaseFunctionIndex
    c%name = name
    c%zLevelBase = zLevelBase
    c%horizontallyUniform = uniform
    c%table = copy_PhaseFunctionTable(table)
  end subroutine fillComponent

  subroutine releaseComponent(c)
    type(opticalComponent), intent(inout) :: c
    if(associated(c%extinction))             deallocate(c%extinction)
    if(associated(c%singleScatteringAlbedo)) deallocate(c%singleScatteringAlbedo)
    if(associated(c%phaseFunctionIndex))     deallocate(c%phaseFunctionIndex)
    call finalize_PhaseFunctionTable(c%table)
    c%name = ""; c%zLevelBase = 0; c%horizontallyUniform = .false.
  end subroutine releaseComponent

  subroutine addComponent3D(thisDomain, componentName, extinction, singleScatteringAlbedo, &
                            phaseFunctionIndex, phaseFunctions, zLevelBase, status)
    type(domain),                intent(inout) :: thisDomain
    character(len = *),          intent(in   ) :: componentName
    real,    dimension(:, :, :), intent(in   ) :: extinction, singleScatteringAlbedo
    integer, dimension(:, :, :), intent(in   ) :: phaseFunctionIndex
    type(phaseFunctionTable),    intent(in   ) :: phaseFunctions
    integer, optional,           intent(in   ) :: zLevelBase
    type(ErrorMessage),          intent(inout) :: status
    call appendComponent(thisDomain, componentName, extinction, singleScatteringAlbedo, phaseFunctionIndex, &
                         phaseFunctions, zLevelBase, .false., status)
  end subroutine addComponent3D

  subroutine addComponent1D(thisDomain, componentName, extinction, singleScatteringAlbedo, &
                            phaseFunctionIndex, phaseFunctions, zLevelBase, status)
    type(domain),             intent(inout) :: thisDomain
    character(len = *),       intent(in   ) :: componentName
    real,    dimension(:),    intent(in   ) :: extinction, singleScatteringAlbedo
    integer, dimension(:),    intent(in   ) :: phaseFunctionIndex
    type(phaseFunctionTable), intent(in   ) :: phaseFunctions
    integer, optional,        intent(in   ) :: zLevelBase
    type(ErrorMessage),       intent(inout) :: status
    if(size(extinction) /= size(singleScatteringAlbedo) .or. size(extinction) /= size(phaseFunctionIndex)) then
      call setStateToFailure(status, "addOpticalComponent: optical property vectors must be the same length.")
      return
    end if
    call appendComponent(thisDomain, componentName, reshape(extinction, (/ 1, 1, size(extinction) /)),           &
                         reshape(singleScatteringAlbedo, (/ 1, 1, size(extinction) /)),                          &
                         reshape(phaseFunctionIndex, (/ 1, 1, size(extinction) /)), phaseFunctions, zLevelBase, .true., status)
  end subroutine addComponent1D

  subroutine appendComponent(thisDomain, componentName, extinction, singleScatteringAlbedo, phaseFunctionIndex, &
                             phaseFunctions, zLevelBase, uniform, status)
    type(domain),                intent(inout) :: thisDomain
    character(len = *),          intent(in   ) :: componentName
    real,    dimension(:, :, :), intent(in   ) :: extinction, singleScatteringAlbedo
    integer, dimension(:, :, :), intent(in   ) :: phaseFunctionIndex
    type(phaseFunctionTable),    intent(in   ) :: phaseFunctions
    integer, optional,           intent(in   ) :: zLevelBase
    logical,                     intent(in   ) :: uniform
    type(ErrorMessage),          intent(inout) :: status
    type(opticalComponent), dimension(:), pointer :: grown
    integer :: base, nOld

    base = 1
    if(present(zLevelBase)) base = zLevelBase
    call validateComponent(thisDomain, extinction, singleScatteringAlbedo, phaseFunctionIndex, phaseFunctions, base, status)
    if(stateIsFailure(status)) then
      call setStateToFailure(status, "addOpticalComponent: optical properties aren't valid.")
      return
    end if
    nOld = 0
    if(associated(thisDomain%components)) nOld = size(thisDomain%components)
    allocate(grown(nOld + 1))
    if(nOld > 0) then
      grown(:nOld) = thisDomain%components(:)      ! shallow: pointer components move to the new array
      deallocate(thisDomain%components)
    end if
    call fillComponent(grown(nOld + 1), componentName, extinction, singleScatteringAlbedo, phaseFunctionIndex, base, &
                       phaseFunctions, uniform)
    thisDomain%components => grown
    call setStateToSuccess(status)
  end subroutine appendComponent

  subroutine replaceComponent3D(thisDomain, componentNumber, componentName, extinction, singleScatteringAlbedo, &
                                phaseFunctionIndex, phaseFunctions, zLevelBase, status)
    type(domain),                intent(inout) :: thisDomain
    integer,                     intent(in   ) :: componentNumber
    character(len = *),          intent(in   ) :: componentName
    real,    dimension(:, :, :), intent(in   ) :: extinction, singleScatteringAlbedo
    integer, dimension(:, :, :), intent(in   ) :: phaseFunctionIndex
    type(phaseFunctionTable),    intent(in   ) :: phaseFunctions
    integer, optional,           intent(in   ) :: zLevelBase
    type(ErrorMessage),          intent(inout) :: status
    call swapComponent(thisDomain, componentNumber, componentName, extinction, singleScatteringAlbedo, &
                       phaseFunctionIndex, phaseFunctions, zLevelBase, .false., status)
  end subroutine replaceComponent3D

  subroutine replaceComponent1D(thisDomain, componentNumber, componentName, extinction, singleScatteringAlbedo, &
                                phaseFunctionIndex, phaseFunctions, zLevelBase, status)
    type(domain),             intent(inout) :: thisDomain
    integer,                  intent(in   ) :: componentNumber
    character(len = *),       intent(in   ) :: componentName
    real,    dimension(:),    intent(in   ) :: extinction, singleScatteringAlbedo
    integer, dimension(:),    intent(in   ) :: phaseFunctionIndex
    type(phaseFunctionTable), intent(in   ) :: phaseFunctions
    integer, optional,        intent(in   ) :: zLevelBase
    type(ErrorMessage),       intent(inout) :: status
    call swapComponent(thisDomain, componentNumber, componentName, reshape(extinction, (/ 1, 1, size(extinction) /)), &
                       reshape(singleScatteringAlbedo, (/ 1, 1, size(extinction) /)),                                &
                       reshape(phaseFunctionIndex, (/ 1, 1, size(extinction) /)), phaseFunctions, zLevelBase, .true., status)
  end subroutine replaceComponent1D

  subroutine swapComponent(thisDomain, componentNumber, componentName, extinction, singleScatteringAlbedo, &
                           phaseFunctionIndex, phaseFunctions, zLevelBase, uniform, status)
    type(domain),                intent(inout) :: thisDomain
    integer,                     intent(in   ) :: componentNumber
    character(len = *),          intent(in   ) :: componentName
    real,    dimension(:, :, :), intent(in   ) :: extinction, singleScatteringAlbedo
    integer, dimension(:, :, :), intent(in   ) :: phaseFunctionIndex
    type(phaseFunctionTable),    intent(in   ) :: phaseFunctions
    integer, optional,           intent(in   ) :: zLevelBase
    logical,                     intent(in   ) :: uniform
    type(ErrorMessage),          intent(inout) :: status
    integer :: base
    base = 1
    if(present(zLevelBase)) base = zLevelBase
    if(.not. associated(thisDomain%components)) then
      call setStateToFailure(status, "replaceOpticalComponent: no components to replace.")
      return
    end if
    if(componentNumber < 1 .or. componentNumber > size(thisDomain%components)) then
      call setStateToFailure(status, "replaceOpticalComponent: no components to replace.")
      return
    end if
    call validateComponent(thisDomain, extinction, singleScatteringAlbedo, phaseFunctionIndex, phaseFunctions, base, status)
    if(stateIsFailure(status)) then
      call setStateToFailure(status, "replaceOpticalComponent: optical properties aren't valid.")
      return
    end if
    call releaseComponent(thisDomain%components(componentNumber))
    call fillComponent(thisDomain%components(componentNumber), componentName, extinction, singleScatteringAlbedo, &
                       phaseFunctionIndex, base, phaseFunctions, uniform)
    call setStateToSuccess(status)
  end subroutine swapComponent

  subroutine deleteOpticalComponent(thisDomain, componentNumber, status)
    type(domain),       intent(inout) :: thisDomain
    integer,            intent(in   ) :: componentNumber
    type(ErrorMessage), intent(inout) :: status
    type(opticalComponent), dimension(:), pointer :: shrunk
    integer :: n
    n = 0
    if(associated(thisDomain%components)) n = size(thisDomain%components)
    if(componentNumber < 1 .or. componentNumber > n) then
      call setStateToFailure(status, "deleteOpticalComponent: no such component.")
      return
    end if
    call releaseComponent(thisDomain%components(componentNumber))
    if(n == 1) then
      deallocate(thisDomain%components); nullify(thisDomain%components)
    else
      allocate(shrunk(n - 1))
      shrunk(:componentNumber - 1) = thisDomain%components(:componentNumber - 1)
      shrunk(componentNumber:)     = thisDomain%components(componentNumber + 1:)
      deallocate(thisDomain%components)
      thisDomain%components => shrunk
    end if
    call setStateToSuccess(status)
  end subroutine deleteOpticalComponent

  ! ------------------------------------------------------------------------------------------------
  ! Inquiry
  ! ------------------------------------------------------------------------------------------------
  subroutine getInfo_Domain(thisDomain, numX, numY, numZ, xPosition, yPosition, zPosition, &
                            numberOfComponents, componentNames, status)
    type(domain),                               intent(in   ) :: thisDomain
    integer,                          optional, intent(  out) :: numX, numY, numZ
    real,    dimension(:),            optional, intent(  out) :: xPosition, yPosition, zPosition
    integer,                          optional, intent(  out) :: numberOfComponents
    character(len = *), dimension(:), optional, intent(  out) :: componentNames
    type(ErrorMessage),                         intent(inout) :: status
    integer :: n

    if(.not. isValid(thisDomain)) then
      call setStateToFailure(status, "getInfo_Domain: domain hasn't been initialized.")
      return
    end if
    if(present(numX)) numX = size(thisDomain%xPosition) - 1
    if(present(numY)) numY = size(thisDomain%yPosition) - 1
    if(present(numZ)) numZ = size(thisDomain%zPosition) - 1
    if(present(xPosition)) call copyEdges(thisDomain%xPosition, xPosition, "x")
    if(present(yPosition)) call copyEdges(thisDomain%yPosition, yPosition, "y")
    if(present(zPosition)) call copyEdges(thisDomain%zPosition, zPosition, "z")
    n = 0
    if(associated(thisDomain%components)) n = size(thisDomain%components)
    if(present(numberOfComponents)) numberOfComponents = n
    if(present(componentNames)) then
      componentNames(:) = ""
      if(size(componentNames) < n) then
        call setStateToFailure(status, "getInfo_Domain: component names array is wrong length")
      else if(n > 0) then
        componentNames(:n) = thisDomain%components(:)%name
      end if
    end if
    if(.not. stateIsFailure(status)) call setStateToSuccess(status)
  contains
    subroutine copyEdges(from, to, axis)
      real, dimension(:), intent(in ) :: from
      real, dimension(:), intent(out) :: to
      character(len = 1), intent(in ) :: axis
      if(size(to) /= size(from)) then
        call setStateToFailure(status, "getInfo_Domain: vector for " // axis // " positions is wrong length.")
      else
        to(:) = from(:)
      end if
    end subroutine copyEdges
  end subroutine getInfo_Domain

  ! Properties expanded to the full grid, component by component: totalExtinction(x,y,z); for component c the
  ! cumulative fraction of the extinction due to components 1..c, its single scattering albedo and its
  ! phase-function index, all (x,y,z,c); zero where a component is absent.
  subroutine getOpticalPropertiesByComponent(thisDomain, totalExtinction, cumulativeExtinction, singleScatteringAlbedo, &
                                             phaseFunctionIndex, phaseFunctions, status)
    type(domain),                                     intent(in   ) :: thisDomain
    real,    dimension(:, :, :),                      intent(  out) :: totalExtinction
    real,    dimension(:, :, :, :),                   intent(  out) :: cumulativeExtinction, singleScatteringAlbedo
    integer, dimension(:, :, :, :),                   intent(  out) :: phaseFunctionIndex
    type(phaseFunctionTable), dimension(:), optional, intent(  out) :: phaseFunctions
    type(ErrorMessage),                               intent(inout) :: status
    integer :: nx, ny, nz, nc, c, k, k0, k1, i, j

    if(.not. isValid(thisDomain)) then
      call setStateToFailure(status, "getOpticalPropertiesByComponent: domain is not initialized.")
      return
    else if(.not. associated(thisDomain%components)) then
      call setStateToFailure(status, "getOpticalPropertiesByComponent: domain contains no optical components.")
      return
    end if
    nx = size(thisDomain%xPosition) - 1; ny = size(thisDomain%yPosition) - 1; nz = size(thisDomain%zPosition) - 1
    nc = size(thisDomain%components)
    if(any(shape(totalExtinction) /= (/ nx, ny, nz /)) .or. any(shape(cumulativeExtinction) /= (/ nx, ny, nz, nc /)) .or. &
       any(shape(singleScatteringAlbedo) /= (/ nx, ny, nz, nc /)) .or. any(shape(phaseFunctionIndex) /= (/ nx, ny, nz, nc /))) &
      call setStateToFailure(status, "getOpticalPropertiesByComponent: extent of one or more arrays incorrect.")
    if(present(phaseFunctions)) then
      if(size(phaseFunctions) /= nc) &
        call setStateToFailure(status, "getOpticalPropertiesByComponent: number of components in phaseFunctions array.")
    end if
    if(stateIsFailure(status)) return

    cumulativeExtinction = 0.; singleScatteringAlbedo = 0.; phaseFunctionIndex = 0
    do c = 1, nc
      k0 = thisDomain%components(c)%zLevelBase
      k1 = k0 + size(thisDomain%components(c)%extinction, 3) - 1
      if(thisDomain%components(c)%horizontallyUniform) then
        do k = k0, k1
          cumulativeExtinction  (:, :, k, c) = thisDomain%components(c)%extinction(1, 1, k - k0 + 1)
          singleScatteringAlbedo(:, :, k, c) = thisDomain%components(c)%singleScatteringAlbedo(1, 1, k - k0 + 1)
          phaseFunctionIndex    (:, :, k, c) = thisDomain%components(c)%phaseFunctionIndex(1, 1, k - k0 + 1)
        end do
      else
        cumulativeExtinction  (:, :, k0:k1, c) = thisDomain%components(c)%extinction
        singleScatteringAlbedo(:, :, k0:k1, c) = thisDomain%components(c)%singleScatteringAlbedo
        phaseFunctionIndex    (:, :, k0:k1, c) = thisDomain%components(c)%phaseFunctionIndex
      end if
      if(present(phaseFunctions)) phaseFunctions(c) = copy_PhaseFunctionTable(thisDomain%components(c)%table)
    end do
    do c = 2, nc
      cumulativeExtinction(:, :, :, c) = cumulativeExtinction(:, :, :, c) + cumulativeExtinction(:, :, :, c - 1)
    end do
    totalExtinction(:, :, :) = cumulativeExtinction(:, :, :, nc)
    do k = 1, nz
      do j = 1, ny
        do i = 1, nx
          if(totalExtinction(i, j, k) > tiny(totalExtinction)) &
            cumulativeExtinction(i, j, k, :) = cumulativeExtinction(i, j, k, :) / totalExtinction(i, j, k)
        end do
      end do
    end do
    call setStateToSuccess(status)
  end subroutine getOpticalPropertiesByComponent

  ! ------------------------------------------------------------------------------------------------
  ! Files (netCDF classic), schema of Code/opticalProperties.f95:554-844
  ! ------------------------------------------------------------------------------------------------
  function componentPrefix(i) result(prefix)
    integer, intent(in) :: i
    character(len = 32) :: prefix
    prefix = "Component" // trim(IntToChar(i)) // "_"
  end function componentPrefix

  subroutine write_Domain(thisDomain, fileName, status)
    type(domain),       intent(in   ) :: thisDomain
    character(len = *), intent(in   ) :: fileName
    type(ErrorMessage), intent(inout) :: status
    integer :: rc(12), ncid, xE, yE, zE, xG, yG, zG, zDim, v, c, nc
    integer(kind = selected_int_kind(2)) :: flagXY, flagZ   ! (one-byte attributes, as the reference's asInt makes them, :593-594, :1018-1022:
                                                            !  its own write_Domain over this netCDF module gives this writer's bytes)
    integer, dimension(:), allocatable :: extVar, ssaVar, idxVar

    if(.not. isValid(thisDomain)) then
      call setStateToFailure(status, "write_Domain: domain hasn't been initialized.")
      return
    end if
    nc = 0
    if(associated(thisDomain%components)) nc = size(thisDomain%components)
    flagXY = 0; flagZ = 0
    if(thisDomain%xyRegularlySpaced) flagXY = 1
    if(thisDomain%zRegularlySpaced)  flagZ  = 1
    rc(:) = nf90_noerr
    rc(1)  = nf90_create(trim(fileName), nf90_clobber, ncid)
    if(rc(1) /= nf90_noerr) then
      call setStateToFailure(status, "write_Domain: can't create file " // trim(fileName))
      return
    end if
    rc(2)  = nf90_def_dim(ncid, "x-Edges", size(thisDomain%xPosition), xE)
    rc(3)  = nf90_def_dim(ncid, "y-Edges", size(thisDomain%yPosition), yE)
    rc(4)  = nf90_def_dim(ncid, "z-Edges", size(thisDomain%zPosition), zE)
    rc(5)  = nf90_def_dim(ncid, "x-Grid",  size(thisDomain%xPosition) - 1, xG)
    rc(6)  = nf90_def_dim(ncid, "y-Grid",  size(thisDomain%yPosition) - 1, yG)
    rc(7)  = nf90_def_dim(ncid, "z-Grid",  size(thisDomain%zPosition) - 1, zG)
    rc(8)  = nf90_def_var(ncid, "x-Edges", nf90_float, xE, v)
    rc(9)  = nf90_def_var(ncid, "y-Edges", nf90_float, yE, v)
    rc(10) = nf90_def_var(ncid, "z-Edges", nf90_float, zE, v)
    rc(11) = nf90_put_att(ncid, nf90_global, "xyRegularlySpaced", flagXY)
    rc(12) = nf90_put_att(ncid, nf90_global, "zRegularlySpaced",  flagZ)
    if(any(rc /= nf90_noerr)) call setStateToFailure(status, "write_Domain: error writing domain information")
    allocate(extVar(max(nc, 1)), ssaVar(max(nc, 1)), idxVar(max(nc, 1)))
    if(.not. stateIsFailure(status) .and. nc > 0) then
      rc(1) = nf90_put_att(ncid, nf90_global, "numberOfComponents", nc)
      do c = 1, nc
        rc(:) = nf90_noerr
        rc(1) = nf90_put_att(ncid, nf90_global, trim(componentPrefix(c)) // "Name", trim(thisDomain%components(c)%name))
        rc(2) = nf90_put_att(ncid, nf90_global, trim(componentPrefix(c)) // "zLevelBase", thisDomain%components(c)%zLevelBase)
        if(thisDomain%components(c)%zLevelBase == 1 .and. &
           size(thisDomain%components(c)%extinction, 3) == size(thisDomain%zPosition) - 1) then
          zDim = zG
        else
          rc(3) = nf90_def_dim(ncid, trim(componentPrefix(c)) // "z-Grid", size(thisDomain%components(c)%extinction, 3), zDim)
        end if
        if(thisDomain%components(c)%horizontallyUniform) then
          rc(4) = nf90_def_var(ncid, trim(componentPrefix(c)) // "Extinction",             nf90_float, zDim, extVar(c))
          rc(5) = nf90_def_var(ncid, trim(componentPrefix(c)) // "SingleScatteringAlbedo", nf90_float, zDim, ssaVar(c))
          rc(6) = nf90_def_var(ncid, trim(componentPrefix(c)) // "PhaseFunctionIndex",     nf90_short, zDim, idxVar(c))
        else
          rc(4) = nf90_def_var(ncid, trim(componentPrefix(c)) // "Extinction",             nf90_float, (/ xG, yG, zDim /), extVar(c))
          rc(5) = nf90_def_var(ncid, trim(componentPrefix(c)) // "SingleScatteringAlbedo", nf90_float, (/ xG, yG, zDim /), ssaVar(c))
          rc(6) = nf90_def_var(ncid, trim(componentPrefix(c)) // "PhaseFunctionIndex",     nf90_short, (/ xG, yG, zDim /), idxVar(c))
        end if
        if(any(rc /= nf90_noerr)) &
          call setStateToFailure(status, "write_Domain: Error creating definitions for component" // trim(IntToChar(c)))
      end do
    end if
    rc(:) = nf90_noerr
    rc(1) = nf90_enddef(ncid)
    rc(2) = nf90_inq_varid(ncid, "x-Edges", v); rc(3) = nf90_put_var(ncid, v, thisDomain%xPosition)
    rc(4) = nf90_inq_varid(ncid, "y-Edges", v); rc(5) = nf90_put_var(ncid, v, thisDomain%yPosition)
    rc(6) = nf90_inq_varid(ncid, "z-Edges", v); rc(7) = nf90_put_var(ncid, v, thisDomain%zPosition)
    if(any(rc /= nf90_noerr)) call setStateToFailure(status, "write_Domain: error writing domain data")
    do c = 1, nc
      if(stateIsFailure(status)) exit
      rc(:) = nf90_noerr
      if(thisDomain%components(c)%horizontallyUniform) then
        rc(1) = nf90_put_var(ncid, extVar(c), thisDomain%components(c)%extinction(1, 1, :))
        rc(2) = nf90_put_var(ncid, ssaVar(c), thisDomain%components(c)%singleScatteringAlbedo(1, 1, :))
        rc(3) = nf90_put_var(ncid, idxVar(c), thisDomain%components(c)%phaseFunctionIndex(1, 1, :))
      else
        rc(1) = nf90_put_var(ncid, extVar(c), thisDomain%components(c)%extinction)
        rc(2) = nf90_put_var(ncid, ssaVar(c), thisDomain%components(c)%singleScatteringAlbedo)
        rc(3) = nf90_put_var(ncid, idxVar(c), thisDomain%components(c)%phaseFunctionIndex)
      end if
      if(any(rc /= nf90_noerr)) call setStateToFailure(status, "write_Domain: " // trim(nf90_strerror(minval(rc))))
      call add_PhaseFunctionTable(thisDomain%components(c)%table, fileId = ncid, prefix = trim(componentPrefix(c)), status = status)
    end do
    deallocate(extVar, ssaVar, idxVar)
    rc(1) = nf90_close(ncid)
    if(stateIsFailure(status)) then
      open(20, file = trim(fileName)); close(20, status = "delete")
    else
      call setStateToSuccess(status)
    end if
  end subroutine write_Domain

  subroutine read_Domain(fileName, thisDomain, status)
    character(len = *), intent(in   ) :: fileName
    type(domain),       intent(  out) :: thisDomain
    type(ErrorMessage), intent(inout) :: status
    integer :: rc(12), ncid, dimId, varId, nxE, nyE, nzE, nc, c, zBase, nzc, nd, dimIds(3)
    character(len = nameLength) :: name
    real,    dimension(:),       allocatable :: xE, yE, zE
    real,    dimension(:, :, :), allocatable :: ext, ssa
    integer, dimension(:, :, :), allocatable :: idx
    type(phaseFunctionTable) :: table
    logical :: uniform

    rc(:) = nf90_noerr
    if(nf90_open(trim(fileName), nf90_nowrite, ncid) /= nf90_noerr) then
      call setStateToFailure(status, "read_Domain: Can't open file " // trim(fileName))
      return
    end if
    rc(1) = nf90_inq_dimid(ncid, "x-Edges", dimId); if(rc(1) == 0) rc(2) = nf90_inquire_dimension(ncid, dimId, len = nxE)
    rc(3) = nf90_inq_dimid(ncid, "y-Edges", dimId); if(rc(3) == 0) rc(4) = nf90_inquire_dimension(ncid, dimId, len = nyE)
    rc(5) = nf90_inq_dimid(ncid, "z-Edges", dimId); if(rc(5) == 0) rc(6) = nf90_inquire_dimension(ncid, dimId, len = nzE)
    if(any(rc /= nf90_noerr)) then
      call setStateToFailure(status, "read_Domain: " // trim(fileName) // " doesn't look an optical properties file.")
      rc(1) = nf90_close(ncid)
      return
    end if
    allocate(xE(nxE), yE(nyE), zE(nzE))
    rc(1) = nf90_inq_varid(ncid, "x-Edges", varId); if(rc(1) == 0) rc(2) = nf90_get_var(ncid, varId, xE)
    rc(3) = nf90_inq_varid(ncid, "y-Edges", varId); if(rc(3) == 0) rc(4) = nf90_get_var(ncid, varId, yE)
    rc(5) = nf90_inq_varid(ncid, "z-Edges", varId); if(rc(5) == 0) rc(6) = nf90_get_var(ncid, varId, zE)
    if(any(rc /= nf90_noerr)) then
      call setStateToFailure(status, "read_Domain: " // trim(nf90_strerror(minval(rc))))
    else
      thisDomain = new_Domain(xE, yE, zE, status)
    end if
    nc = 0
    if(.not. stateIsFailure(status)) then
      if(nf90_get_att(ncid, nf90_global, "numberOfComponents", nc) /= nf90_noerr) nc = 0
    end if
    do c = 1, nc
      if(stateIsFailure(status)) exit
      rc(:) = nf90_noerr
      rc(1) = nf90_get_att(ncid, nf90_global, trim(componentPrefix(c)) // "Name", name)
      rc(2) = nf90_get_att(ncid, nf90_global, trim(componentPrefix(c)) // "zLevelBase", zBase)
      rc(3) = nf90_inq_varid(ncid, trim(componentPrefix(c)) // "Extinction", varId)
      if(rc(3) == nf90_noerr) rc(4) = nf90_inquire_variable(ncid, varId, ndims = nd, dimids = dimIds)
      if(any(rc /= nf90_noerr)) then
        call setStateToFailure(status, "read_Domain: Error reading component " // trim(IntToChar(c)))
        exit
      end if
      uniform = nd == 1
      rc(5) = nf90_inquire_dimension(ncid, dimIds(nd), len = nzc)
      if(uniform) then
        allocate(ext(1, 1, nzc), ssa(1, 1, nzc), idx(1, 1, nzc))
        rc(6) = nf90_get_var(ncid, varId, ext(1, 1, :))
        rc(7) = nf90_inq_varid(ncid, trim(componentPrefix(c)) // "SingleScatteringAlbedo", varId)
        if(rc(7) == 0) rc(8) = nf90_get_var(ncid, varId, ssa(1, 1, :))
        rc(9) = nf90_inq_varid(ncid, trim(componentPrefix(c)) // "PhaseFunctionIndex", varId)
        if(rc(9) == 0) rc(10) = nf90_get_var(ncid, varId, idx(1, 1, :))
      else
        allocate(ext(nxE - 1, nyE - 1, nzc), ssa(nxE - 1, nyE - 1, nzc), idx(nxE - 1, nyE - 1, nzc))
        rc(6) = nf90_get_var(ncid, varId, ext)
        rc(7) = nf90_inq_varid(ncid, trim(componentPrefix(c)) // "SingleScatteringAlbedo", varId)
        if(rc(7) == 0) rc(8) = nf90_get_var(ncid, varId, ssa)
        rc(9) = nf90_inq_varid(ncid, trim(componentPrefix(c)) // "PhaseFunctionIndex", varId)
        if(rc(9) == 0) rc(10) = nf90_get_var(ncid, varId, idx)
      end if
      if(any(rc /= nf90_noerr)) then
        call setStateToFailure(status, "read_Domain: " // trim(nf90_strerror(minval(rc))))
      else
        call read_PhaseFunctionTable(fileId = ncid, table = table, prefix = trim(componentPrefix(c)), status = status)
        if(stateIsFailure(status)) then
          call setStateToFailure(status, "read_Domain: Error reading phase function table.")
        else if(uniform) then
          call addComponent1D(thisDomain, trim(name), ext(1, 1, :), ssa(1, 1, :), idx(1, 1, :), table, zLevelBase = zBase, &
                              status = status)
        else
          call addComponent3D(thisDomain, trim(name), ext, ssa, idx, table, zLevelBase = zBase, status = status)
        end if
        call finalize_PhaseFunctionTable(table)
      end if
      deallocate(ext, ssa, idx)
    end do
    deallocate(xE, yE, zE)
    rc(1) = nf90_close(ncid)
    if(stateIsFailure(status)) then
      call finalize_Domain(thisDomain)
    else
      call setStateToSuccess(status)
    end if
  end subroutine read_Domain

  subroutine finalize_Domain(thisDomain)
    type(domain), intent(inout) :: thisDomain
    integer :: c
    if(associated(thisDomain%xPosition)) deallocate(thisDomain%xPosition)
    if(associated(thisDomain%yPosition)) deallocate(thisDomain%yPosition)
    if(associated(thisDomain%zPosition)) deallocate(thisDomain%zPosition)
    if(associated(thisDomain%components)) then
      do c = 1, size(thisDomain%components)
        call releaseComponent(thisDomain%components(c))
      end do
      deallocate(thisDomain%components)
    end if
    thisDomain%xyRegularlySpaced = .false.; thisDomain%zRegularlySpaced = .false.
  end subroutine finalize_Domain
end module opticalProperties
