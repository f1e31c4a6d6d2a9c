! Minimal netCDF "classic" (CDF-1) reader / writer, exposed as module `netcdf` with the nf90_* names the I3RC code
! uses (domain files: Code/opticalProperties.f95:554-844; phase-function tables:
! Code/scatteringPhaseFunctions.f95:899-1252; results: Example-Drivers/monteCarloDriver.f95:609-854).
!
! Written from the published file-format specification (NetCDF Users Guide, "File Format Specification",
! classic format): magic 'CDF' 1, numrecs, dim_list, gatt_list, var_list, then the fixed-size variables' data,
! everything big-endian and padded to 4 bytes.  Supported: fixed dimensions (no record dimension), global and
! variable attributes of type char / byte / short / int / float / double, variables of type byte / short / int /
! float / double with up to 4 dimensions.  A dataset is held in memory and written on nf90_close.
! This is a from-scratch subset, NOT Unidata's library: no CDF-2/5, HDF5, record variables, or partial access.
module netcdf
  implicit none
  private

  integer, parameter, public :: nf90_noerr = 0, nf90_global = 0
  integer, parameter, public :: nf90_clobber = 0, nf90_noclobber = 4, nf90_nowrite = 0, nf90_write = 1
  integer, parameter, public :: nf90_byte = 1, nf90_char = 2, nf90_short = 3, nf90_int = 4, nf90_float = 5, nf90_double = 6
  integer, parameter, public :: nf90_ebadid = -33, nf90_enotvar = -49, nf90_ebaddim = -46, nf90_enotatt = -43, &
                                nf90_einval = -36, nf90_eedge = -57, nf90_enotnc = -51, nf90_enameinuse = -42, &
                                nf90_ebadtype = -45, nf90_emaxvars = -48, nf90_eperm = -37, nf90_eio = -68
  integer, parameter, public :: nf90_max_name = 256, nf90_max_var_dims = 4

  integer, parameter :: i1 = selected_int_kind(2), i2 = selected_int_kind(4), i4 = selected_int_kind(9), &
                        r4 = kind(1.0), r8 = kind(1.0d0)
  integer, parameter :: maxOpen = 8, maxDims = 64, maxVars = 256, maxAtts = 64, maxAttLen = 1024

  type attribute
    character(len = nf90_max_name) :: name = ""
    integer                        :: xtype = 0, n = 0
    character(len = maxAttLen)     :: text = ""
    real(r8), allocatable          :: values(:)
  end type attribute

  type variable
    character(len = nf90_max_name) :: name = ""
    integer                        :: xtype = 0, ndims = 0, dimids(nf90_max_var_dims) = 0, natts = 0
    type(attribute), allocatable   :: atts(:)
    real(r8), allocatable          :: values(:)        ! column-major order of the nf90 (Fortran) dimension order
  end type variable

  type dataset
    logical                        :: inUse = .false., writable = .false., dirty = .false.
    character(len = 1024)          :: path = ""
    integer                        :: ndims = 0, nvars = 0, ngatts = 0
    character(len = nf90_max_name) :: dimName(maxDims) = ""
    integer                        :: dimLen(maxDims) = 0
    type(attribute), allocatable   :: gatts(:)
    type(variable),  allocatable   :: vars(:)
  end type dataset

  type(dataset), save, target :: files(maxOpen)

  interface nf90_def_var
    module procedure defVarScalar, defVarOneDim, defVarManyDims
  end interface
  interface nf90_put_att
    module procedure putAttText, putAttReal, putAttReals, putAttInt, putAttInts, putAttByte
  end interface
  interface nf90_get_att
    module procedure getAttText, getAttReal, getAttReals, getAttInt, getAttInts, getAttByte
  end interface
  interface nf90_put_var
    module procedure putVarR1, putVarR2, putVarR3, putVarR4, putVarI0, putVarI1, putVarI2, putVarI3, putVarR0
  end interface
  interface nf90_get_var
    module procedure getVarR1, getVarR2, getVarR3, getVarR4, getVarI0, getVarI1, getVarI2, getVarI3, getVarR0
  end interface

  public :: nf90_create, nf90_open, nf90_close, nf90_redef, nf90_enddef, nf90_inquire, nf90_strerror
  public :: nf90_def_dim, nf90_inq_dimid, nf90_inquire_dimension
  public :: nf90_def_var, nf90_inq_varid, nf90_inquire_variable
  public :: nf90_put_att, nf90_get_att, nf90_put_var, nf90_get_var
contains
  ! ================================================================================================
  ! Dataset life cycle
  ! ================================================================================================
  function freeSlot() result(k)
    integer :: k
    do k = 1, maxOpen
      if(.not. files(k)%inUse) return
    end do
    k = 0
  end function freeSlot

  logical function valid(ncid)
    integer, intent(in) :: ncid
    valid = ncid >= 1 .and. ncid <= maxOpen
    if(valid) valid = files(ncid)%inUse
  end function valid

  subroutine reset(f)
    type(dataset), intent(inout) :: f
    if(allocated(f%gatts)) deallocate(f%gatts)
    if(allocated(f%vars))  deallocate(f%vars)
    f%inUse = .false.; f%writable = .false.; f%dirty = .false.
    f%ndims = 0; f%nvars = 0; f%ngatts = 0; f%path = ""
  end subroutine reset

  function nf90_create(path, cmode, ncid) result(rc)
    character(len = *), intent(in ) :: path
    integer,            intent(in ) :: cmode
    integer,            intent(out) :: ncid
    integer :: rc, unit, ios
    ncid = freeSlot()
    if(ncid == 0) then
      rc = nf90_emaxvars; return
    end if
    open(newunit = unit, file = trim(path), access = "stream", form = "unformatted", status = "replace", iostat = ios)
    if(ios /= 0) then
      rc = nf90_eio; ncid = 0; return
    end if
    close(unit)
    call reset(files(ncid))
    allocate(files(ncid)%gatts(maxAtts), files(ncid)%vars(maxVars))
    files(ncid)%inUse = .true.; files(ncid)%writable = .true.; files(ncid)%dirty = .true.
    files(ncid)%path = path
    rc = nf90_noerr
  end function nf90_create

  function nf90_redef(ncid) result(rc)
    integer, intent(in) :: ncid
    integer :: rc
    rc = nf90_noerr
    if(.not. valid(ncid)) rc = nf90_ebadid
  end function nf90_redef

  function nf90_enddef(ncid) result(rc)
    integer, intent(in) :: ncid
    integer :: rc
    rc = nf90_noerr
    if(.not. valid(ncid)) rc = nf90_ebadid
  end function nf90_enddef

  function nf90_inquire(ncid, nDimensions, nVariables, nAttributes) result(rc)
    integer,           intent(in ) :: ncid
    integer, optional, intent(out) :: nDimensions, nVariables, nAttributes
    integer :: rc
    if(.not. valid(ncid)) then
      rc = nf90_ebadid; return
    end if
    if(present(nDimensions)) nDimensions = files(ncid)%ndims
    if(present(nVariables))  nVariables  = files(ncid)%nvars
    if(present(nAttributes)) nAttributes = files(ncid)%ngatts
    rc = nf90_noerr
  end function nf90_inquire

  function nf90_strerror(rc) result(text)
    integer, intent(in) :: rc
    character(len = 80) :: text
    select case(rc)
      case(nf90_noerr);      text = "No error"
      case(nf90_ebadid);     text = "NetCDF: Not a valid ID"
      case(nf90_enotvar);    text = "NetCDF: Variable not found"
      case(nf90_ebaddim);    text = "NetCDF: Invalid dimension ID or name"
      case(nf90_enotatt);    text = "NetCDF: Attribute not found"
      case(nf90_einval);     text = "NetCDF: Invalid argument"
      case(nf90_eedge);      text = "NetCDF: Start+count exceeds dimension bound"
      case(nf90_enotnc);     text = "NetCDF: Unknown file format (only the classic CDF-1 format is supported)"
      case(nf90_enameinuse); text = "NetCDF: String match to name in use"
      case(nf90_ebadtype);   text = "NetCDF: Not a valid data type or unsupported here"
      case(nf90_emaxvars);   text = "NetCDF: table of open files / variables / attributes is full"
      case(nf90_eperm);      text = "NetCDF: Write to read only"
      case(nf90_eio);        text = "NetCDF: file could not be opened, read or written"
      case default;          text = "NetCDF: unknown error"
    end select
  end function nf90_strerror

  ! ================================================================================================
  ! Dimensions
  ! ================================================================================================
  function nf90_def_dim(ncid, name, len, dimid) result(rc)
    integer,            intent(in ) :: ncid, len
    character(len = *), intent(in ) :: name
    integer,            intent(out) :: dimid
    integer :: rc, k
    dimid = 0
    if(.not. valid(ncid)) then
      rc = nf90_ebadid; return
    end if
    if(len < 1) then
      rc = nf90_einval; return          ! no record (unlimited) dimension in this subset
    end if
    do k = 1, files(ncid)%ndims
      if(trim(files(ncid)%dimName(k)) == trim(name)) then
        rc = nf90_enameinuse; return
      end if
    end do
    if(files(ncid)%ndims >= maxDims) then
      rc = nf90_emaxvars; return
    end if
    files(ncid)%ndims = files(ncid)%ndims + 1
    dimid = files(ncid)%ndims
    files(ncid)%dimName(dimid) = name
    files(ncid)%dimLen(dimid)  = len
    files(ncid)%dirty = .true.
    rc = nf90_noerr
  end function nf90_def_dim

  function nf90_inq_dimid(ncid, name, dimid) result(rc)
    integer,            intent(in ) :: ncid
    character(len = *), intent(in ) :: name
    integer,            intent(out) :: dimid
    integer :: rc, k
    dimid = 0
    rc = nf90_ebadid
    if(.not. valid(ncid)) return
    rc = nf90_ebaddim
    do k = 1, files(ncid)%ndims
      if(trim(files(ncid)%dimName(k)) == trim(name)) then
        dimid = k; rc = nf90_noerr; return
      end if
    end do
  end function nf90_inq_dimid

  function nf90_inquire_dimension(ncid, dimid, name, len) result(rc)
    integer,                      intent(in ) :: ncid, dimid
    character(len = *), optional, intent(out) :: name
    integer,            optional, intent(out) :: len
    integer :: rc
    rc = nf90_ebadid
    if(.not. valid(ncid)) return
    rc = nf90_ebaddim
    if(dimid < 1 .or. dimid > files(ncid)%ndims) return
    if(present(name)) name = files(ncid)%dimName(dimid)
    if(present(len))  len  = files(ncid)%dimLen(dimid)
    rc = nf90_noerr
  end function nf90_inquire_dimension

  ! ================================================================================================
  ! Variables
  ! ================================================================================================
  function defVarManyDims(ncid, name, xtype, dimids, varid) result(rc)
    integer,               intent(in ) :: ncid, xtype
    character(len = *),    intent(in ) :: name
    integer, dimension(:), intent(in ) :: dimids
    integer,               intent(out) :: varid
    integer :: rc, k, total
    varid = 0
    rc = nf90_ebadid
    if(.not. valid(ncid)) return
    rc = nf90_ebadtype
    if(xtype /= nf90_byte .and. xtype /= nf90_short .and. xtype /= nf90_int .and. xtype /= nf90_float .and. &
       xtype /= nf90_double) return
    rc = nf90_einval
    if(size(dimids) > nf90_max_var_dims) return
    rc = nf90_ebaddim
    if(any(dimids < 1) .or. any(dimids > files(ncid)%ndims)) return
    rc = nf90_enameinuse
    do k = 1, files(ncid)%nvars
      if(trim(files(ncid)%vars(k)%name) == trim(name)) return
    end do
    rc = nf90_emaxvars
    if(files(ncid)%nvars >= maxVars) return
    files(ncid)%nvars = files(ncid)%nvars + 1
    varid = files(ncid)%nvars
    files(ncid)%vars(varid)%name   = name
    files(ncid)%vars(varid)%xtype  = xtype
    files(ncid)%vars(varid)%ndims  = size(dimids)
    files(ncid)%vars(varid)%dimids(:size(dimids)) = dimids
    files(ncid)%vars(varid)%natts  = 0
    total = 1
    do k = 1, size(dimids)
      total = total * files(ncid)%dimLen(dimids(k))
    end do
    if(allocated(files(ncid)%vars(varid)%values)) deallocate(files(ncid)%vars(varid)%values)
    allocate(files(ncid)%vars(varid)%values(total))
    files(ncid)%vars(varid)%values(:) = 0._r8
    if(.not. allocated(files(ncid)%vars(varid)%atts)) allocate(files(ncid)%vars(varid)%atts(maxAtts))
    files(ncid)%dirty = .true.
    rc = nf90_noerr
  end function defVarManyDims

  function defVarOneDim(ncid, name, xtype, dimids, varid) result(rc)
    integer,            intent(in ) :: ncid, xtype, dimids
    character(len = *), intent(in ) :: name
    integer,            intent(out) :: varid
    integer :: rc
    rc = defVarManyDims(ncid, name, xtype, (/ dimids /), varid)
  end function defVarOneDim

  function defVarScalar(ncid, name, xtype, varid) result(rc)
    integer,            intent(in ) :: ncid, xtype
    character(len = *), intent(in ) :: name
    integer,            intent(out) :: varid
    integer :: rc
    integer :: none(0)
    rc = defVarManyDims(ncid, name, xtype, none, varid)
  end function defVarScalar

  function nf90_inq_varid(ncid, name, varid) result(rc)
    integer,            intent(in ) :: ncid
    character(len = *), intent(in ) :: name
    integer,            intent(out) :: varid
    integer :: rc, k
    varid = 0
    rc = nf90_ebadid
    if(.not. valid(ncid)) return
    rc = nf90_enotvar
    do k = 1, files(ncid)%nvars
      if(trim(files(ncid)%vars(k)%name) == trim(name)) then
        varid = k; rc = nf90_noerr; return
      end if
    end do
  end function nf90_inq_varid

  function nf90_inquire_variable(ncid, varid, name, xtype, ndims, dimids, nAtts) result(rc)
    integer,                         intent(in ) :: ncid, varid
    character(len = *),    optional, intent(out) :: name
    integer,               optional, intent(out) :: xtype, ndims, nAtts
    integer, dimension(:), optional, intent(out) :: dimids
    integer :: rc, n
    rc = nf90_ebadid
    if(.not. valid(ncid)) return
    rc = nf90_enotvar
    if(varid < 1 .or. varid > files(ncid)%nvars) return
    n = files(ncid)%vars(varid)%ndims
    if(present(name))  name  = files(ncid)%vars(varid)%name
    if(present(xtype)) xtype = files(ncid)%vars(varid)%xtype
    if(present(ndims)) ndims = n
    if(present(nAtts)) nAtts = files(ncid)%vars(varid)%natts
    if(present(dimids)) dimids(:min(n, size(dimids))) = files(ncid)%vars(varid)%dimids(:min(n, size(dimids)))
    rc = nf90_noerr
  end function nf90_inquire_variable

  ! -- data access: whole variables only ---------------------------------------------------------------
  function storeValues(ncid, varid, flat) result(rc)
    integer,  intent(in) :: ncid, varid
    real(r8), intent(in) :: flat(:)
    integer :: rc
    rc = nf90_ebadid
    if(.not. valid(ncid)) return
    rc = nf90_eperm
    if(.not. files(ncid)%writable) return
    rc = nf90_enotvar
    if(varid < 1 .or. varid > files(ncid)%nvars) return
    rc = nf90_eedge
    if(size(flat) /= size(files(ncid)%vars(varid)%values)) return
    files(ncid)%vars(varid)%values(:) = flat(:)
    files(ncid)%dirty = .true.
    rc = nf90_noerr
  end function storeValues

  function loadValues(ncid, varid, flat) result(rc)
    integer,  intent(in ) :: ncid, varid
    real(r8), intent(out) :: flat(:)
    integer :: rc
    rc = nf90_ebadid
    if(.not. valid(ncid)) return
    rc = nf90_enotvar
    if(varid < 1 .or. varid > files(ncid)%nvars) return
    rc = nf90_eedge
    if(size(flat) /= size(files(ncid)%vars(varid)%values)) return
    flat(:) = files(ncid)%vars(varid)%values(:)
    rc = nf90_noerr
  end function loadValues

  function putVarR0(ncid, varid, values) result(rc)
    integer, intent(in) :: ncid, varid
    real,    intent(in) :: values
    integer :: rc
    rc = storeValues(ncid, varid, (/ real(values, r8) /))
  end function putVarR0
  function putVarR1(ncid, varid, values) result(rc)
    integer, intent(in) :: ncid, varid
    real,    intent(in) :: values(:)
    integer :: rc
    rc = storeValues(ncid, varid, real(values, r8))
  end function putVarR1
  function putVarR2(ncid, varid, values) result(rc)
    integer, intent(in) :: ncid, varid
    real,    intent(in) :: values(:, :)
    integer :: rc
    rc = storeValues(ncid, varid, real(reshape(values, (/ size(values) /)), r8))
  end function putVarR2
  function putVarR3(ncid, varid, values) result(rc)
    integer, intent(in) :: ncid, varid
    real,    intent(in) :: values(:, :, :)
    integer :: rc
    rc = storeValues(ncid, varid, real(reshape(values, (/ size(values) /)), r8))
  end function putVarR3
  function putVarR4(ncid, varid, values) result(rc)
    integer, intent(in) :: ncid, varid
    real,    intent(in) :: values(:, :, :, :)
    integer :: rc
    rc = storeValues(ncid, varid, real(reshape(values, (/ size(values) /)), r8))
  end function putVarR4
  function putVarI0(ncid, varid, values) result(rc)
    integer, intent(in) :: ncid, varid
    integer, intent(in) :: values
    integer :: rc
    rc = storeValues(ncid, varid, (/ real(values, r8) /))
  end function putVarI0
  function putVarI1(ncid, varid, values) result(rc)
    integer, intent(in) :: ncid, varid
    integer, intent(in) :: values(:)
    integer :: rc
    rc = storeValues(ncid, varid, real(values, r8))
  end function putVarI1
  function putVarI2(ncid, varid, values) result(rc)
    integer, intent(in) :: ncid, varid
    integer, intent(in) :: values(:, :)
    integer :: rc
    rc = storeValues(ncid, varid, real(reshape(values, (/ size(values) /)), r8))
  end function putVarI2
  function putVarI3(ncid, varid, values) result(rc)
    integer, intent(in) :: ncid, varid
    integer, intent(in) :: values(:, :, :)
    integer :: rc
    rc = storeValues(ncid, varid, real(reshape(values, (/ size(values) /)), r8))
  end function putVarI3

  function getVarR0(ncid, varid, values) result(rc)
    integer, intent(in ) :: ncid, varid
    real,    intent(out) :: values
    integer :: rc
    real(r8) :: flat(1)
    rc = loadValues(ncid, varid, flat)
    if(rc == nf90_noerr) values = real(flat(1))
  end function getVarR0
  function getVarR1(ncid, varid, values) result(rc)
    integer, intent(in ) :: ncid, varid
    real,    intent(out) :: values(:)
    integer :: rc
    real(r8), allocatable :: flat(:)
    allocate(flat(size(values)))
    rc = loadValues(ncid, varid, flat)
    if(rc == nf90_noerr) values = real(flat)
    deallocate(flat)
  end function getVarR1
  function getVarR2(ncid, varid, values) result(rc)
    integer, intent(in ) :: ncid, varid
    real,    intent(out) :: values(:, :)
    integer :: rc
    real(r8), allocatable :: flat(:)
    allocate(flat(size(values)))
    rc = loadValues(ncid, varid, flat)
    if(rc == nf90_noerr) values = reshape(real(flat), shape(values))
    deallocate(flat)
  end function getVarR2
  function getVarR3(ncid, varid, values) result(rc)
    integer, intent(in ) :: ncid, varid
    real,    intent(out) :: values(:, :, :)
    integer :: rc
    real(r8), allocatable :: flat(:)
    allocate(flat(size(values)))
    rc = loadValues(ncid, varid, flat)
    if(rc == nf90_noerr) values = reshape(real(flat), shape(values))
    deallocate(flat)
  end function getVarR3
  function getVarR4(ncid, varid, values) result(rc)
    integer, intent(in ) :: ncid, varid
    real,    intent(out) :: values(:, :, :, :)
    integer :: rc
    real(r8), allocatable :: flat(:)
    allocate(flat(size(values)))
    rc = loadValues(ncid, varid, flat)
    if(rc == nf90_noerr) values = reshape(real(flat), shape(values))
    deallocate(flat)
  end function getVarR4
  function getVarI0(ncid, varid, values) result(rc)
    integer, intent(in ) :: ncid, varid
    integer, intent(out) :: values
    integer :: rc
    real(r8) :: flat(1)
    rc = loadValues(ncid, varid, flat)
    if(rc == nf90_noerr) values = nint(flat(1))
  end function getVarI0
  function getVarI1(ncid, varid, values) result(rc)
    integer, intent(in ) :: ncid, varid
    integer, intent(out) :: values(:)
    integer :: rc
    real(r8), allocatable :: flat(:)
    allocate(flat(size(values)))
    rc = loadValues(ncid, varid, flat)
    if(rc == nf90_noerr) values = nint(flat)
    deallocate(flat)
  end function getVarI1
  function getVarI2(ncid, varid, values) result(rc)
    integer, intent(in ) :: ncid, varid
    integer, intent(out) :: values(:, :)
    integer :: rc
    real(r8), allocatable :: flat(:)
    allocate(flat(size(values)))
    rc = loadValues(ncid, varid, flat)
    if(rc == nf90_noerr) values = reshape(nint(flat), shape(values))
    deallocate(flat)
  end function getVarI2
  function getVarI3(ncid, varid, values) result(rc)
    integer, intent(in ) :: ncid, varid
    integer, intent(out) :: values(:, :, :)
    integer :: rc
    real(r8), allocatable :: flat(:)
    allocate(flat(size(values)))
    rc = loadValues(ncid, varid, flat)
    if(rc == nf90_noerr) values = reshape(nint(flat), shape(values))
    deallocate(flat)
  end function getVarI3

  ! ================================================================================================
  ! Attributes
  ! ================================================================================================
  ! slot of attribute `name` of (ncid, varid); create = append if missing.  0 if absent / no room.
  function attSlot(ncid, varid, name, create) result(k)
    integer,            intent(in) :: ncid, varid
    character(len = *), intent(in) :: name
    logical,            intent(in) :: create
    integer :: k, n
    k = 0
    if(varid == nf90_global) then
      n = files(ncid)%ngatts
      do k = 1, n
        if(trim(files(ncid)%gatts(k)%name) == trim(name)) return
      end do
      k = 0
      if(create .and. n < maxAtts) then
        files(ncid)%ngatts = n + 1; k = n + 1
        files(ncid)%gatts(k)%name = name
      end if
    else
      n = files(ncid)%vars(varid)%natts
      do k = 1, n
        if(trim(files(ncid)%vars(varid)%atts(k)%name) == trim(name)) return
      end do
      k = 0
      if(create .and. n < maxAtts) then
        files(ncid)%vars(varid)%natts = n + 1; k = n + 1
        files(ncid)%vars(varid)%atts(k)%name = name
      end if
    end if
  end function attSlot

  function putAttGeneric(ncid, varid, name, xtype, text, values) result(rc)
    integer,            intent(in) :: ncid, varid, xtype
    character(len = *), intent(in) :: name
    character(len = *), intent(in), optional :: text
    real(r8),           intent(in), optional :: values(:)
    integer :: rc, k
    rc = nf90_ebadid
    if(.not. valid(ncid)) return
    rc = nf90_eperm
    if(.not. files(ncid)%writable) return
    rc = nf90_enotvar
    if(varid /= nf90_global .and. (varid < 1 .or. varid > files(ncid)%nvars)) return
    k = attSlot(ncid, varid, name, .true.)
    rc = nf90_emaxvars
    if(k == 0) return
    if(varid == nf90_global) then
      call fill(files(ncid)%gatts(k))
    else
      call fill(files(ncid)%vars(varid)%atts(k))
    end if
    files(ncid)%dirty = .true.
    rc = nf90_noerr
  contains
    subroutine fill(a)
      type(attribute), intent(inout) :: a
      a%xtype = xtype
      if(allocated(a%values)) deallocate(a%values)
      if(present(text)) then
        a%text = text
        a%n = min(len(text), maxAttLen)
      else
        allocate(a%values(size(values)))
        a%values(:) = values(:)
        a%n = size(values)
      end if
    end subroutine fill
  end function putAttGeneric

  function putAttText(ncid, varid, name, values) result(rc)
    integer,            intent(in) :: ncid, varid
    character(len = *), intent(in) :: name, values
    integer :: rc
    rc = putAttGeneric(ncid, varid, name, nf90_char, text = values)
  end function putAttText
  function putAttReal(ncid, varid, name, values) result(rc)
    integer,            intent(in) :: ncid, varid
    character(len = *), intent(in) :: name
    real,               intent(in) :: values
    integer :: rc
    rc = putAttGeneric(ncid, varid, name, nf90_float, values = (/ real(values, r8) /))
  end function putAttReal
  function putAttReals(ncid, varid, name, values) result(rc)
    integer,            intent(in) :: ncid, varid
    character(len = *), intent(in) :: name
    real,               intent(in) :: values(:)
    integer :: rc
    rc = putAttGeneric(ncid, varid, name, nf90_float, values = real(values, r8))
  end function putAttReals
  function putAttInt(ncid, varid, name, values) result(rc)
    integer,            intent(in) :: ncid, varid
    character(len = *), intent(in) :: name
    integer,            intent(in) :: values
    integer :: rc
    rc = putAttGeneric(ncid, varid, name, nf90_int, values = (/ real(values, r8) /))
  end function putAttInt
  function putAttInts(ncid, varid, name, values) result(rc)
    integer,            intent(in) :: ncid, varid
    character(len = *), intent(in) :: name
    integer,            intent(in) :: values(:)
    integer :: rc
    rc = putAttGeneric(ncid, varid, name, nf90_int, values = real(values, r8))
  end function putAttInts
  function putAttByte(ncid, varid, name, values) result(rc)
    integer,            intent(in) :: ncid, varid
    character(len = *), intent(in) :: name
    integer(i1),        intent(in) :: values
    integer :: rc
    rc = putAttGeneric(ncid, varid, name, nf90_byte, values = (/ real(values, r8) /))
  end function putAttByte

  ! numeric attribute values converted to float64; rc /= 0 if missing or textual
  function numericAtt(ncid, varid, name, vals, n) result(rc)
    integer,            intent(in ) :: ncid, varid
    character(len = *), intent(in ) :: name
    real(r8),           intent(out) :: vals(:)
    integer,            intent(out) :: n
    integer :: rc, k
    n = 0
    rc = nf90_ebadid
    if(.not. valid(ncid)) return
    rc = nf90_enotvar
    if(varid /= nf90_global .and. (varid < 1 .or. varid > files(ncid)%nvars)) return
    k = attSlot(ncid, varid, name, .false.)
    rc = nf90_enotatt
    if(k == 0) return
    if(varid == nf90_global) then
      call take(files(ncid)%gatts(k))
    else
      call take(files(ncid)%vars(varid)%atts(k))
    end if
  contains
    subroutine take(a)
      type(attribute), intent(in) :: a
      rc = nf90_ebadtype
      if(a%xtype == nf90_char) return
      n = min(a%n, size(vals))
      vals(:n) = a%values(:n)
      rc = nf90_noerr
    end subroutine take
  end function numericAtt

  function getAttText(ncid, varid, name, values) result(rc)
    integer,            intent(in ) :: ncid, varid
    character(len = *), intent(in ) :: name
    character(len = *), intent(out) :: values
    integer :: rc, k
    values = ""
    rc = nf90_ebadid
    if(.not. valid(ncid)) return
    rc = nf90_enotvar
    if(varid /= nf90_global .and. (varid < 1 .or. varid > files(ncid)%nvars)) return
    k = attSlot(ncid, varid, name, .false.)
    rc = nf90_enotatt
    if(k == 0) return
    rc = nf90_ebadtype
    if(varid == nf90_global) then
      if(files(ncid)%gatts(k)%xtype /= nf90_char) return
      values = files(ncid)%gatts(k)%text(:files(ncid)%gatts(k)%n)
    else
      if(files(ncid)%vars(varid)%atts(k)%xtype /= nf90_char) return
      values = files(ncid)%vars(varid)%atts(k)%text(:files(ncid)%vars(varid)%atts(k)%n)
    end if
    rc = nf90_noerr
  end function getAttText
  function getAttReal(ncid, varid, name, values) result(rc)
    integer,            intent(in ) :: ncid, varid
    character(len = *), intent(in ) :: name
    real,               intent(out) :: values
    integer :: rc, n
    real(r8) :: v(1)
    rc = numericAtt(ncid, varid, name, v, n)
    if(rc == nf90_noerr) values = real(v(1))
  end function getAttReal
  function getAttReals(ncid, varid, name, values) result(rc)
    integer,            intent(in ) :: ncid, varid
    character(len = *), intent(in ) :: name
    real,               intent(out) :: values(:)
    integer :: rc, n
    real(r8), allocatable :: v(:)
    allocate(v(size(values)))
    rc = numericAtt(ncid, varid, name, v, n)
    if(rc == nf90_noerr) values(:n) = real(v(:n))
    deallocate(v)
  end function getAttReals
  function getAttInt(ncid, varid, name, values) result(rc)
    integer,            intent(in ) :: ncid, varid
    character(len = *), intent(in ) :: name
    integer,            intent(out) :: values
    integer :: rc, n
    real(r8) :: v(1)
    rc = numericAtt(ncid, varid, name, v, n)
    if(rc == nf90_noerr) values = nint(v(1))
  end function getAttInt
  function getAttInts(ncid, varid, name, values) result(rc)
    integer,            intent(in ) :: ncid, varid
    character(len = *), intent(in ) :: name
    integer,            intent(out) :: values(:)
    integer :: rc, n
    real(r8), allocatable :: v(:)
    allocate(v(size(values)))
    rc = numericAtt(ncid, varid, name, v, n)
    if(rc == nf90_noerr) values(:n) = nint(v(:n))
    deallocate(v)
  end function getAttInts
  function getAttByte(ncid, varid, name, values) result(rc)
    integer,            intent(in ) :: ncid, varid
    character(len = *), intent(in ) :: name
    integer(i1),        intent(out) :: values
    integer :: rc, n
    real(r8) :: v(1)
    rc = numericAtt(ncid, varid, name, v, n)
    if(rc == nf90_noerr) values = int(nint(v(1)), i1)
  end function getAttByte

  ! ================================================================================================
  ! File image: big-endian encode / decode
  ! ================================================================================================
  pure function typeSize(xtype) result(n)
    integer, intent(in) :: xtype
    integer :: n
    select case(xtype)
      case(nf90_byte, nf90_char); n = 1
      case(nf90_short);           n = 2
      case(nf90_int, nf90_float); n = 4
      case default;               n = 8
    end select
  end function typeSize

  pure function padded(n) result(m)
    integer, intent(in) :: n
    integer :: m
    m = 4 * ((n + 3) / 4)
  end function padded

  subroutine putInt(buf, pos, v)      ! 32-bit big-endian
    integer(i1), intent(inout) :: buf(:)
    integer,     intent(inout) :: pos
    integer,     intent(in   ) :: v
    integer :: b
    do b = 0, 3
      buf(pos + b) = int(ibits(v, 8 * (3 - b), 8) - merge(256, 0, ibits(v, 8 * (3 - b), 8) > 127), i1)
    end do
    pos = pos + 4
  end subroutine putInt

  function getInt(buf, pos) result(v)
    integer(i1), intent(in   ) :: buf(:)
    integer,     intent(inout) :: pos
    integer :: v, b
    v = 0
    do b = 0, 3
      v = ior(ishft(v, 8), iand(int(buf(pos + b)), 255))
    end do
    pos = pos + 4
  end function getInt

  subroutine putName(buf, pos, name)
    integer(i1),        intent(inout) :: buf(:)
    integer,            intent(inout) :: pos
    character(len = *), intent(in   ) :: name
    integer :: n, k
    n = len_trim(name)
    call putInt(buf, pos, n)
    do k = 1, n
      buf(pos + k - 1) = int(iachar(name(k:k)), i1)
    end do
    pos = pos + padded(n)
  end subroutine putName

  subroutine getName(buf, pos, name)
    integer(i1),        intent(in   ) :: buf(:)
    integer,            intent(inout) :: pos
    character(len = *), intent(  out) :: name
    integer :: n, k
    n = getInt(buf, pos)
    name = ""
    do k = 1, min(n, len(name))
      name(k:k) = achar(iand(int(buf(pos + k - 1)), 255))
    end do
    pos = pos + padded(n)
  end subroutine getName

  ! n values of external type xtype from / to float64, advancing pos (no padding here)
  subroutine putValues(buf, pos, xtype, vals)
    integer(i1), intent(inout) :: buf(:)
    integer,     intent(inout) :: pos
    integer,     intent(in   ) :: xtype
    real(r8),    intent(in   ) :: vals(:)
    integer :: k, b, iv
    integer(i1) :: raw4(4), raw8(8)
    do k = 1, size(vals)
      select case(xtype)
        case(nf90_byte)
          buf(pos) = int(nint(vals(k)), i1); pos = pos + 1
        case(nf90_short)
          iv = nint(vals(k))
          buf(pos)     = int(ibits(iv, 8, 8) - merge(256, 0, ibits(iv, 8, 8) > 127), i1)
          buf(pos + 1) = int(ibits(iv, 0, 8) - merge(256, 0, ibits(iv, 0, 8) > 127), i1)
          pos = pos + 2
        case(nf90_int)
          call putInt(buf, pos, nint(vals(k)))
        case(nf90_float)
          raw4 = transfer(real(vals(k), r4), raw4)     ! little-endian host
          do b = 1, 4
            buf(pos + b - 1) = raw4(5 - b)
          end do
          pos = pos + 4
        case default
          raw8 = transfer(vals(k), raw8)
          do b = 1, 8
            buf(pos + b - 1) = raw8(9 - b)
          end do
          pos = pos + 8
      end select
    end do
  end subroutine putValues

  subroutine getValues(buf, pos, xtype, vals)
    integer(i1), intent(in   ) :: buf(:)
    integer,     intent(inout) :: pos
    integer,     intent(in   ) :: xtype
    real(r8),    intent(  out) :: vals(:)
    integer :: k, b, iv
    integer(i1) :: raw4(4), raw8(8)
    real(r4) :: f
    do k = 1, size(vals)
      select case(xtype)
        case(nf90_byte)
          vals(k) = real(buf(pos), r8); pos = pos + 1
        case(nf90_short)
          iv = ior(ishft(iand(int(buf(pos)), 255), 8), iand(int(buf(pos + 1)), 255))
          if(iv > 32767) iv = iv - 65536
          vals(k) = real(iv, r8); pos = pos + 2
        case(nf90_int)
          vals(k) = real(getInt(buf, pos), r8)
        case(nf90_float)
          do b = 1, 4
            raw4(5 - b) = buf(pos + b - 1)
          end do
          f = transfer(raw4, f)
          vals(k) = real(f, r8); pos = pos + 4
        case default
          do b = 1, 8
            raw8(9 - b) = buf(pos + b - 1)
          end do
          vals(k) = transfer(raw8, vals(k)); pos = pos + 8
      end select
    end do
  end subroutine getValues

  pure function attListBytes(atts, n) result(total)
    type(attribute), intent(in) :: atts(:)
    integer,         intent(in) :: n
    integer :: total, k
    total = 8
    do k = 1, n
      total = total + 4 + padded(len_trim(atts(k)%name)) + 8 + padded(atts(k)%n * typeSize(atts(k)%xtype))
    end do
  end function attListBytes

  subroutine putAttList(buf, pos, atts, n)
    integer(i1),     intent(inout) :: buf(:)
    integer,         intent(inout) :: pos
    type(attribute), intent(in   ) :: atts(:)
    integer,         intent(in   ) :: n
    integer :: k, c, start
    if(n == 0) then
      call putInt(buf, pos, 0); call putInt(buf, pos, 0)
      return
    end if
    call putInt(buf, pos, 12)            ! NC_ATTRIBUTE
    call putInt(buf, pos, n)
    do k = 1, n
      call putName(buf, pos, atts(k)%name)
      call putInt(buf, pos, atts(k)%xtype)
      call putInt(buf, pos, atts(k)%n)
      start = pos
      if(atts(k)%xtype == nf90_char) then
        do c = 1, atts(k)%n
          buf(pos + c - 1) = int(iachar(atts(k)%text(c:c)), i1)
        end do
        pos = pos + atts(k)%n
      else
        call putValues(buf, pos, atts(k)%xtype, atts(k)%values(:atts(k)%n))
      end if
      pos = start + padded(pos - start)
    end do
  end subroutine putAttList

  subroutine getAttList(buf, pos, atts, n, rc)
    integer(i1),     intent(in   ) :: buf(:)
    integer,         intent(inout) :: pos
    type(attribute), intent(inout) :: atts(:)
    integer,         intent(  out) :: n, rc
    integer :: tag, k, c, start, count
    rc = nf90_noerr
    tag = getInt(buf, pos)
    count = getInt(buf, pos)
    n = 0
    if(tag == 0 .and. count == 0) return
    if(tag /= 12 .or. count > size(atts)) then
      rc = nf90_enotnc; return
    end if
    n = count
    do k = 1, n
      call getName(buf, pos, atts(k)%name)
      atts(k)%xtype = getInt(buf, pos)
      atts(k)%n     = getInt(buf, pos)
      start = pos
      if(allocated(atts(k)%values)) deallocate(atts(k)%values)
      if(atts(k)%xtype == nf90_char) then
        atts(k)%text = ""
        do c = 1, min(atts(k)%n, maxAttLen)
          atts(k)%text(c:c) = achar(iand(int(buf(pos + c - 1)), 255))
        end do
        atts(k)%n = min(atts(k)%n, maxAttLen)
        pos = start + padded(getCount(buf, start))      ! true stored length, even if truncated above
      else
        allocate(atts(k)%values(atts(k)%n))
        call getValues(buf, pos, atts(k)%xtype, atts(k)%values)
        pos = start + padded(pos - start)
      end if
    end do
  contains
    integer function getCount(b, at)
      integer(i1), intent(in) :: b(:)
      integer,     intent(in) :: at
      integer :: p
      p = at - 4
      getCount = getInt(b, p)
    end function getCount
  end subroutine getAttList

  ! ================================================================================================
  ! open / close
  ! ================================================================================================
  function nf90_close(ncid) result(rc)
    integer, intent(in) :: ncid
    integer :: rc, total, headerBytes, pos, k, d, unit, ios, nbytes
    integer(i1), allocatable :: buf(:)
    integer, allocatable :: begin(:), vsize(:)
    type(dataset), pointer :: f

    rc = nf90_ebadid
    if(.not. valid(ncid)) return
    f => files(ncid)
    rc = nf90_noerr
    if(f%writable) then
      ! ---- sizes
      headerBytes = 4 + 4                                   ! magic, numrecs
      headerBytes = headerBytes + 8
      do k = 1, f%ndims
        headerBytes = headerBytes + 4 + padded(len_trim(f%dimName(k))) + 4
      end do
      headerBytes = headerBytes + attListBytes(f%gatts, f%ngatts)
      headerBytes = headerBytes + 8
      allocate(begin(max(f%nvars, 1)), vsize(max(f%nvars, 1)))
      do k = 1, f%nvars
        headerBytes = headerBytes + 4 + padded(len_trim(f%vars(k)%name)) + 4 + 4 * f%vars(k)%ndims + &
                      attListBytes(f%vars(k)%atts, f%vars(k)%natts) + 4 + 4 + 4
        vsize(k) = padded(size(f%vars(k)%values) * typeSize(f%vars(k)%xtype))
      end do
      total = headerBytes
      do k = 1, f%nvars
        begin(k) = total
        total = total + vsize(k)
      end do
      allocate(buf(total))
      buf(:) = 0_i1
      ! ---- header
      buf(1) = int(iachar("C"), i1); buf(2) = int(iachar("D"), i1); buf(3) = int(iachar("F"), i1); buf(4) = 1_i1
      pos = 5
      call putInt(buf, pos, 0)                              ! numrecs
      if(f%ndims == 0) then
        call putInt(buf, pos, 0); call putInt(buf, pos, 0)
      else
        call putInt(buf, pos, 10); call putInt(buf, pos, f%ndims)      ! NC_DIMENSION
        do k = 1, f%ndims
          call putName(buf, pos, f%dimName(k)); call putInt(buf, pos, f%dimLen(k))
        end do
      end if
      call putAttList(buf, pos, f%gatts, f%ngatts)
      if(f%nvars == 0) then
        call putInt(buf, pos, 0); call putInt(buf, pos, 0)
      else
        call putInt(buf, pos, 11); call putInt(buf, pos, f%nvars)      ! NC_VARIABLE
        do k = 1, f%nvars
          call putName(buf, pos, f%vars(k)%name)
          call putInt(buf, pos, f%vars(k)%ndims)
          do d = f%vars(k)%ndims, 1, -1          ! file order is slowest first = reverse of the nf90 order
            call putInt(buf, pos, f%vars(k)%dimids(d) - 1)
          end do
          call putAttList(buf, pos, f%vars(k)%atts, f%vars(k)%natts)
          call putInt(buf, pos, f%vars(k)%xtype)
          call putInt(buf, pos, vsize(k))
          call putInt(buf, pos, begin(k))
        end do
      end if
      if(pos - 1 /= headerBytes) rc = nf90_einval
      ! ---- data
      do k = 1, f%nvars
        pos = begin(k) + 1
        call putValues(buf, pos, f%vars(k)%xtype, f%vars(k)%values)
      end do
      open(newunit = unit, file = trim(f%path), access = "stream", form = "unformatted", status = "replace", iostat = ios)
      if(ios /= 0) then
        rc = nf90_eio
      else
        nbytes = total
        write(unit, iostat = ios) buf(:nbytes)
        if(ios /= 0) rc = nf90_eio
        close(unit)
      end if
      deallocate(buf, begin, vsize)
    end if
    call reset(f)
  end function nf90_close

  function nf90_open(path, mode, ncid) result(rc)
    character(len = *), intent(in ) :: path
    integer,            intent(in ) :: mode
    integer,            intent(out) :: ncid
    integer :: rc, unit, ios, nbytes, pos, tag, count, k, d, total, begin, vs, nd
    integer(i1), allocatable :: buf(:)
    type(dataset), pointer :: f
    integer :: fileDims(nf90_max_var_dims)

    ncid = freeSlot()
    if(ncid == 0) then
      rc = nf90_emaxvars; return
    end if
    rc = nf90_eio
    inquire(file = trim(path), size = nbytes, iostat = ios)
    if(ios /= 0 .or. nbytes < 32) then
      ncid = 0; return
    end if
    open(newunit = unit, file = trim(path), access = "stream", form = "unformatted", status = "old", action = "read", iostat = ios)
    if(ios /= 0) then
      ncid = 0; return
    end if
    allocate(buf(nbytes))
    read(unit, iostat = ios) buf
    close(unit)
    if(ios /= 0) then
      deallocate(buf); ncid = 0; return
    end if
    rc = nf90_enotnc
    if(buf(1) /= int(iachar("C"), i1) .or. buf(2) /= int(iachar("D"), i1) .or. buf(3) /= int(iachar("F"), i1) .or. &
       buf(4) /= 1_i1) then
      deallocate(buf); ncid = 0; return
    end if
    f => files(ncid)
    call reset(f)
    allocate(f%gatts(maxAtts), f%vars(maxVars))
    f%inUse = .true.; f%writable = (mode == nf90_write); f%path = path
    pos = 5
    count = getInt(buf, pos)                                 ! numrecs (must be 0: no record variables)
    tag = getInt(buf, pos); count = getInt(buf, pos)
    if(.not. ((tag == 0 .and. count == 0) .or. tag == 10) .or. count > maxDims) goto 900
    f%ndims = count
    do k = 1, f%ndims
      call getName(buf, pos, f%dimName(k))
      f%dimLen(k) = getInt(buf, pos)
      if(f%dimLen(k) == 0) goto 900                          ! record dimension: unsupported
    end do
    call getAttList(buf, pos, f%gatts, f%ngatts, rc)
    if(rc /= nf90_noerr) goto 900
    rc = nf90_enotnc
    tag = getInt(buf, pos); count = getInt(buf, pos)
    if(.not. ((tag == 0 .and. count == 0) .or. tag == 11) .or. count > maxVars) goto 900
    f%nvars = count
    do k = 1, f%nvars
      call getName(buf, pos, f%vars(k)%name)
      nd = getInt(buf, pos)
      if(nd > nf90_max_var_dims) goto 900
      f%vars(k)%ndims = nd
      do d = 1, nd
        fileDims(d) = getInt(buf, pos) + 1
      end do
      do d = 1, nd
        f%vars(k)%dimids(d) = fileDims(nd + 1 - d)
      end do
      allocate(f%vars(k)%atts(maxAtts))
      call getAttList(buf, pos, f%vars(k)%atts, f%vars(k)%natts, rc)
      if(rc /= nf90_noerr) goto 900
      rc = nf90_enotnc
      f%vars(k)%xtype = getInt(buf, pos)
      vs    = getInt(buf, pos)
      begin = getInt(buf, pos)
      if(f%vars(k)%xtype == nf90_char) goto 900              ! character variables: unsupported
      total = 1
      do d = 1, nd
        total = total * f%dimLen(f%vars(k)%dimids(d))
      end do
      if(begin < 0 .or. begin + total * typeSize(f%vars(k)%xtype) > nbytes) goto 900
      allocate(f%vars(k)%values(total))
      d = begin + 1
      call getValues(buf, d, f%vars(k)%xtype, f%vars(k)%values)
    end do
    deallocate(buf)
    rc = nf90_noerr
    return
900 continue
    deallocate(buf)
    call reset(f)
    ncid = 0
  end function nf90_open
end module netcdf
