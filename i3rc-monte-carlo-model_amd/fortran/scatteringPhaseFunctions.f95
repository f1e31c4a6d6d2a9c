! Fortran-95 shell of the MI355X photon-tracing integrator -- phase functions and tables of them.
! Public interface of the reference's module scatteringPhaseFunctions (Code/scatteringPhaseFunctions.f95:84-95):
! a phase function is held either as Legendre coefficients (l = 1.., P0 = 1 implied) or as angle/value pairs
! (normalised to integrate to 2 over cos(angle) on construction); a table is a keyed list of them.
! File I/O (read_/write_/add_PhaseFunctionTable) goes through module netcdf (netcdf_classic.f95).
module scatteringPhaseFunctions
  use ErrorMessages,    only: ErrorMessage, stateIsFailure, setStateToFailure, setStateToWarning, setStateToSuccess
  use CharacterUtils,   only: IntToChar
  use numericUtilities, only: findIndex, computeLobattoTerms, computeLegendrePolynomials
  use netcdf
  implicit none
  private

  real,    parameter :: Pi = 3.141592654
  real,    parameter :: smallestAngle = 0., largestAngle = Pi
  integer, parameter :: entryTextLength = 64, tableTextLength = 1024

  type phaseFunction
    private
    real, dimension(:), pointer :: scatteringAngle      => null()
    real, dimension(:), pointer :: value                => null()
    real, dimension(:), pointer :: legendreCoefficients => null()
    real                        :: extinction = 0., singleScatteringAlbedo = 0.
    character(len = entryTextLength) :: description = ""
  end type phaseFunction

  type phaseFunctionTable
    private
    type(phaseFunction), dimension(:), pointer :: phaseFunctions => null()
    real,                dimension(:), pointer :: key            => null()
    character(len = tableTextLength)           :: description = ""
    logical                                    :: oneAngleSet = .false.
  end type phaseFunctionTable

  interface new_PhaseFunction
    module procedure phaseFunctionFromPairs, phaseFunctionFromMoments
  end interface
  interface new_PhaseFunctionTable
    module procedure tableFromSharedAngles, tableFromPhaseFunctions
  end interface
  interface getPhaseFunctionValues
    module procedure valuesOfOne, valuesOfTable
  end interface

  public :: phaseFunction, phaseFunctionTable
  public :: new_PhaseFunction,       new_PhaseFunctionTable,      &
            copy_PhaseFunction,      copy_PhaseFunctionTable,     &
            getInfo_PhaseFunction,   getInfo_PhaseFunctionTable,  &
            read_PhaseFunctionTable, add_PhaseFunctionTable,      &
            write_PhaseFunctionTable,                             &
            finalize_PhaseFunction,  finalize_PhaseFunctionTable, &
            isReady_PhaseFunction,   isReady_PhaseFunctionTable,  &
            getElement, getExtinction, getSingleScatteringAlbedo, &
            getPhaseFunctionValues, getPhaseFunctionCoefficients
contains
  ! ------------------------------------------------------------------------------------------------
  ! Constructors
  ! ------------------------------------------------------------------------------------------------
  subroutine checkAngleGrid(who, angles, status)
    character(len = *), intent(in   ) :: who
    real, dimension(:), intent(in   ) :: angles
    type(ErrorMessage), intent(inout) :: status
    integer :: n
    n = size(angles)
    if(any(angles < smallestAngle) .or. any(angles > largestAngle)) &
      call setStateToFailure(status, who // ": ScatteringAngle out of bounds.")
    if(abs(angles(1) - smallestAngle) > spacing(smallestAngle)) &
      call setStateToFailure(status, who // ": First scattering angle must be min value")
    if(abs(angles(n) - largestAngle) > spacing(largestAngle)) &
      call setStateToFailure(status, who // ": Last scattering angle must be max value")
    if(any(angles(2:) - angles(:n - 1) <= 0.)) &
      call setStateToFailure(status, who // ": Scattering angle must be increasing, unique.")
  end subroutine checkAngleGrid

  ! scale so that the trapezoid integral over cos(angle) is 2
  pure function normalized(angles, values)
    real, dimension(:), intent(in) :: angles, values
    real, dimension(size(values))  :: normalized
    integer :: n
    n = size(angles)
    normalized(:) = -values(:) * 2. / dot_product(cos(angles(2:)) - cos(angles(:n - 1)), 0.5 * (values(2:) + values(:n - 1)))
  end function normalized

  function phaseFunctionFromPairs(scatteringAngle, value, extinction, singleScatteringAlbedo, description, status) &
           result(newPhaseFunction)
    real, dimension(:),           intent(in   ) :: value, scatteringAngle
    real,               optional, intent(in   ) :: extinction, singleScatteringAlbedo
    character(len = *), optional, intent(in   ) :: description
    type(ErrorMessage),           intent(inout) :: status
    type(phaseFunction)                         :: newPhaseFunction

    call checkAngleGrid("newPhaseFunction", scatteringAngle, status)
    if(any(value < 0.)) call setStateToFailure(status, "newPhaseFunction: Negative phase function values supplied.")
    if(size(scatteringAngle) /= size(value)) &
      call setStateToFailure(status, "newPhaseFunction: Number of scattering angles and phase function values must match.")
    call checkScalars(extinction, singleScatteringAlbedo, description, status)
    if(stateIsFailure(status)) return
    allocate(newPhaseFunction%scatteringAngle(size(value)), newPhaseFunction%value(size(value)))
    newPhaseFunction%scatteringAngle(:) = scatteringAngle(:)
    newPhaseFunction%value(:)           = normalized(scatteringAngle, value)
    call storeScalars(newPhaseFunction, extinction, singleScatteringAlbedo, description)
    call setStateToSuccess(status)
  end function phaseFunctionFromPairs

  function phaseFunctionFromMoments(legendreCoefficients, extinction, singleScatteringAlbedo, description, status) &
           result(newPhaseFunction)
    real, dimension(:),           intent(in   ) :: legendreCoefficients
    real,               optional, intent(in   ) :: extinction, singleScatteringAlbedo
    character(len = *), optional, intent(in   ) :: description
    type(ErrorMessage),           intent(inout) :: status
    type(phaseFunction)                         :: newPhaseFunction
    integer, parameter :: nProbe = 1801
    real, dimension(nProbe) :: probeAngles, probeValues
    integer :: i, nNegative

    if(size(legendreCoefficients) > 1) then
      if(legendreCoefficients(1) > 1. .or. legendreCoefficients(1) < -1.) &
        call setStateToFailure(status, "newPhaseFunction: Asymmetery parameter out of bounds.")
      if(abs(legendreCoefficients(1) - 1.) < spacing(1.)) &
        call setStateToWarning(status, "newPhaseFunction: Should the first legendre moment (P1) really be 1?")
    end if
    call checkScalars(extinction, singleScatteringAlbedo, description, status)
    if(stateIsFailure(status)) return
    allocate(newPhaseFunction%legendreCoefficients(size(legendreCoefficients)))
    newPhaseFunction%legendreCoefficients(:) = legendreCoefficients(:)
    call storeScalars(newPhaseFunction, extinction, singleScatteringAlbedo, description)
    ! the reference probes the expansion for negative values and only warns
    probeAngles(:) = (/ (i, i = 0, nProbe - 1) /) / real(nProbe - 1) * Pi
    call valuesOfOne(newPhaseFunction, probeAngles, probeValues, status)
    nNegative = count(probeValues < 0.)
    if(nNegative > 0) then
      call setStateToWarning(status, "newPhaseFunction: Phase function coefficients give " // &
                             trim(IntToChar((100 * nNegative) / nProbe)) // "% negative phase function values")
    else
      call setStateToSuccess(status)
    end if
  end function phaseFunctionFromMoments

  subroutine checkScalars(extinction, singleScatteringAlbedo, description, status)
    real,               optional, intent(in   ) :: extinction, singleScatteringAlbedo
    character(len = *), optional, intent(in   ) :: description
    type(ErrorMessage),           intent(inout) :: status
    if(present(extinction)) then
      if(extinction < 0.) call setStateToFailure(status, "newPhaseFunction: negative extinction supplied.")
    end if
    if(present(singleScatteringAlbedo)) then
      if(singleScatteringAlbedo < 0. .or. singleScatteringAlbedo > 1.) &
        call setStateToFailure(status, "newPhaseFunction: singleScatteringAlbedo out of bounds.")
    end if
    if(present(description)) then
      if(len_trim(description) > entryTextLength) &
        call setStateToWarning(status, "newPhaseFunction: description will be trunctated.")
    end if
  end subroutine checkScalars

  subroutine storeScalars(p, extinction, singleScatteringAlbedo, description)
    type(phaseFunction),          intent(inout) :: p
    real,               optional, intent(in   ) :: extinction, singleScatteringAlbedo
    character(len = *), optional, intent(in   ) :: description
    if(present(extinction))             p%extinction             = extinction
    if(present(singleScatteringAlbedo)) p%singleScatteringAlbedo = singleScatteringAlbedo
    if(present(description))            p%description            = description
  end subroutine storeScalars

  subroutine checkKey(key, nEntries, status)
    real, dimension(:), intent(in   ) :: key
    integer,            intent(in   ) :: nEntries
    type(ErrorMessage), intent(inout) :: status
    if(size(key) /= nEntries) &
      call setStateToFailure(status, "newPhaseFunctionTable: Number of phase functions and key values must match.")
    if(any(key(2:) - key(:size(key) - 1) <= 0.)) &
      call setStateToFailure(status, "newPhaseFunctionTable: Key values must be unique, increasing.")
  end subroutine checkKey

  ! one angle grid shared by all entries; values(nAngles, nEntries)
  function tableFromSharedAngles(scatteringAngle, values, key, extinction, singleScatteringAlbedo, &
                                 phaseFunctionDescriptions, tableDescription, status) result(table)
    real, dimension(:),                       intent(in   ) :: scatteringAngle
    real, dimension(:, :),                    intent(in   ) :: values
    real, dimension(:),                       intent(in   ) :: key
    real, dimension(:),             optional, intent(in   ) :: extinction, singleScatteringAlbedo
    character(len = *), dimension(:), optional, intent(in ) :: phaseFunctionDescriptions
    character(len = *),             optional, intent(in   ) :: tableDescription
    type(ErrorMessage),                       intent(inout) :: status
    type(phaseFunctionTable)                                :: table
    integer :: nAngles, nEntries, i

    nAngles = size(scatteringAngle); nEntries = size(values, 2)
    call checkAngleGrid("newPhaseFunctionTable", scatteringAngle, status)
    if(any(values < 0.)) call setStateToFailure(status, "newPhaseFunctionTable: Negative phase function values supplied.")
    if(size(values, 1) /= nAngles) call setStateToFailure(status, &
      "newPhaseFunctionTable: Number of scattering angles and phase function values must match.")
    call checkKey(key, nEntries, status)
    if(present(extinction)) then
      if(size(extinction) /= nEntries) call setStateToFailure(status, &
        "newPhaseFunctionTable: extinction must be provided for each phase function.")
      if(any(extinction < 0.)) call setStateToFailure(status, "newPhaseFunction: negative extinction supplied.")
    end if
    if(present(singleScatteringAlbedo)) then
      if(size(singleScatteringAlbedo) /= nEntries) call setStateToFailure(status, &
        "newPhaseFunctionTable: single scattering albedo must be provided for each phase function.")
      if(any(singleScatteringAlbedo < 0. .or. singleScatteringAlbedo > 1.)) &
        call setStateToFailure(status, "newPhaseFunction: singleScatteringAlbedo must be > 0, <= 1.")
    end if
    if(present(phaseFunctionDescriptions)) then
      if(size(phaseFunctionDescriptions) /= nEntries) call setStateToFailure(status, &
        "newPhaseFunctionTable: number of descriptions must match number of phase functions.")
    end if
    if(stateIsFailure(status)) return

    allocate(table%phaseFunctions(nEntries), table%key(nEntries))
    allocate(table%phaseFunctions(1)%scatteringAngle(nAngles))
    table%phaseFunctions(1)%scatteringAngle(:) = scatteringAngle(:)
    do i = 1, nEntries
      ! entries 2.. alias the first entry's angle grid, as the reference does (freed once)
      if(i > 1) table%phaseFunctions(i)%scatteringAngle => table%phaseFunctions(1)%scatteringAngle
      allocate(table%phaseFunctions(i)%value(nAngles))
      table%phaseFunctions(i)%value(:) = normalized(scatteringAngle, values(:, i))
      if(present(extinction))                table%phaseFunctions(i)%extinction             = extinction(i)
      if(present(singleScatteringAlbedo))    table%phaseFunctions(i)%singleScatteringAlbedo = singleScatteringAlbedo(i)
      if(present(phaseFunctionDescriptions)) table%phaseFunctions(i)%description            = phaseFunctionDescriptions(i)
    end do
    table%key(:) = key(:)
    if(present(tableDescription)) table%description = tableDescription
    table%oneAngleSet = .true.
    call setStateToSuccess(status)
  end function tableFromSharedAngles

  function tableFromPhaseFunctions(phaseFunctions, key, phaseFunctionDescriptions, tableDescription, status) result(table)
    type(phaseFunction), dimension(:),          intent(in   ) :: phaseFunctions
    real, dimension(:),                         intent(in   ) :: key
    character(len = *), dimension(:), optional, intent(in   ) :: phaseFunctionDescriptions
    character(len = *),               optional, intent(in   ) :: tableDescription
    type(ErrorMessage),                         intent(inout) :: status
    type(phaseFunctionTable)                                  :: table
    integer :: i

    call checkKey(key, size(phaseFunctions), status)
    if(stateIsFailure(status)) return
    allocate(table%phaseFunctions(size(phaseFunctions)), table%key(size(key)))
    do i = 1, size(phaseFunctions)
      table%phaseFunctions(i) = copy_PhaseFunction(phaseFunctions(i))
      if(present(phaseFunctionDescriptions)) table%phaseFunctions(i)%description = phaseFunctionDescriptions(i)
    end do
    table%key(:) = key(:)
    if(present(tableDescription)) table%description = tableDescription
    table%oneAngleSet = .false.
    call setStateToSuccess(status)
  end function tableFromPhaseFunctions

  ! ------------------------------------------------------------------------------------------------
  ! Copies, finalizers, readiness
  ! ------------------------------------------------------------------------------------------------
  function copy_PhaseFunction(original) result(thisCopy)
    type(phaseFunction), intent(in) :: original
    type(phaseFunction)             :: thisCopy
    if(associated(original%legendreCoefficients)) then
      allocate(thisCopy%legendreCoefficients(size(original%legendreCoefficients)))
      thisCopy%legendreCoefficients(:) = original%legendreCoefficients(:)
    end if
    if(associated(original%value)) then
      allocate(thisCopy%value(size(original%value)), thisCopy%scatteringAngle(size(original%scatteringAngle)))
      thisCopy%value(:) = original%value(:)
      thisCopy%scatteringAngle(:) = original%scatteringAngle(:)
    end if
    thisCopy%extinction = original%extinction
    thisCopy%singleScatteringAlbedo = original%singleScatteringAlbedo
    thisCopy%description = original%description
  end function copy_PhaseFunction

  function copy_PhaseFunctionTable(original) result(thisCopy)
    type(phaseFunctionTable), intent(in) :: original
    type(phaseFunctionTable)             :: thisCopy
    integer :: i
    if(.not. associated(original%phaseFunctions)) return
    allocate(thisCopy%phaseFunctions(size(original%phaseFunctions)), thisCopy%key(size(original%key)))
    do i = 1, size(original%phaseFunctions)
      thisCopy%phaseFunctions(i) = copy_PhaseFunction(original%phaseFunctions(i))
    end do
    thisCopy%key(:) = original%key(:)
    thisCopy%description = original%description
    ! a table whose entries share one angle grid stays such a table (only those can be written to files): the copies
    ! of entries 2... give up their own grid and alias the copy of entry 1's, as in the original
    thisCopy%oneAngleSet = original%oneAngleSet
    if(thisCopy%oneAngleSet) then
      do i = 2, size(thisCopy%phaseFunctions)
        if(associated(thisCopy%phaseFunctions(i)%scatteringAngle)) deallocate(thisCopy%phaseFunctions(i)%scatteringAngle)
        thisCopy%phaseFunctions(i)%scatteringAngle => thisCopy%phaseFunctions(1)%scatteringAngle
      end do
    end if
  end function copy_PhaseFunctionTable

  subroutine finalize_PhaseFunction(phaseFunctionVar)
    type(phaseFunction), intent(inout) :: phaseFunctionVar
    if(associated(phaseFunctionVar%scatteringAngle))      deallocate(phaseFunctionVar%scatteringAngle)
    if(associated(phaseFunctionVar%value))                deallocate(phaseFunctionVar%value)
    if(associated(phaseFunctionVar%legendreCoefficients)) deallocate(phaseFunctionVar%legendreCoefficients)
    phaseFunctionVar%extinction = 0.; phaseFunctionVar%singleScatteringAlbedo = 0.; phaseFunctionVar%description = ""
  end subroutine finalize_PhaseFunction

  subroutine finalize_PhaseFunctionTable(table)
    type(phaseFunctionTable), intent(inout) :: table
    integer :: i
    if(associated(table%phaseFunctions)) then
      do i = size(table%phaseFunctions), 1, -1
        if(table%oneAngleSet .and. i > 1) nullify(table%phaseFunctions(i)%scatteringAngle)  ! alias of entry 1
        call finalize_PhaseFunction(table%phaseFunctions(i))
      end do
      deallocate(table%phaseFunctions)
    end if
    if(associated(table%key)) deallocate(table%key)
    table%description = ""; table%oneAngleSet = .false.
  end subroutine finalize_PhaseFunctionTable

  elemental function isReady_PhaseFunction(testPhaseFunction)
    type(phaseFunction), intent(in) :: testPhaseFunction
    logical                         :: isReady_PhaseFunction
    isReady_PhaseFunction = associated(testPhaseFunction%value) .or. associated(testPhaseFunction%legendreCoefficients)
  end function isReady_PhaseFunction

  elemental function isReady_PhaseFunctionTable(table)
    type(phaseFunctionTable), intent(in) :: table
    logical                              :: isReady_PhaseFunctionTable
    isReady_PhaseFunctionTable = associated(table%phaseFunctions)
    if(isReady_PhaseFunctionTable) isReady_PhaseFunctionTable = all(isReady_PhaseFunction(table%phaseFunctions))
  end function isReady_PhaseFunctionTable

  elemental function getExtinction(phaseFunctionVar)
    type(phaseFunction), intent(in) :: phaseFunctionVar
    real                            :: getExtinction
    getExtinction = phaseFunctionVar%extinction
  end function getExtinction

  elemental function getSingleScatteringAlbedo(phaseFunctionVar)
    type(phaseFunction), intent(in) :: phaseFunctionVar
    real                            :: getSingleScatteringAlbedo
    getSingleScatteringAlbedo = phaseFunctionVar%singleScatteringAlbedo
  end function getSingleScatteringAlbedo

  ! ------------------------------------------------------------------------------------------------
  ! Inquiry
  ! ------------------------------------------------------------------------------------------------
  subroutine getInfo_PhaseFunction(phaseFunctionVar, nCoefficients, nAngles, nativeAngles, status)
    type(phaseFunction),          intent(in   ) :: phaseFunctionVar
    integer,            optional, intent(  out) :: nCoefficients, nAngles
    real, dimension(:), optional, intent(  out) :: nativeAngles
    type(ErrorMessage), optional, intent(inout) :: status
    integer :: nStoredAngles, nStoredMoments

    nStoredAngles = 0; nStoredMoments = 0
    if(associated(phaseFunctionVar%value))                nStoredAngles  = size(phaseFunctionVar%value)
    if(associated(phaseFunctionVar%legendreCoefficients)) nStoredMoments = size(phaseFunctionVar%legendreCoefficients)
    if(.not. isReady_PhaseFunction(phaseFunctionVar)) then
      if(present(status)) call setStateToFailure(status, "getInfo_PhaseFunction: phase function hasn't been initialized.")
      return
    end if
    if(present(nCoefficients)) nCoefficients = nStoredMoments
    if(present(nAngles))       nAngles       = nStoredAngles
    if(present(nativeAngles)) then
      if(nStoredAngles == 0) then
        if(present(status)) call setStateToFailure(status, "getInfo_PhaseFunction: phase function has no native angles.")
        return
      else if(size(nativeAngles) < nStoredAngles) then
        if(present(status)) call setStateToFailure(status, "getInfo_PhaseFunction: array for native angles is too small.")
        return
      end if
      nativeAngles(:nStoredAngles) = phaseFunctionVar%scatteringAngle(:)
    end if
    if(present(status)) call setStateToSuccess(status)
  end subroutine getInfo_PhaseFunction

  subroutine getInfo_PhaseFunctionTable(table, nEntries, key, extinction, singleScatteringAlbedo, &
                                        phaseFunctionDescriptions, tableDescription, status)
    type(phaseFunctionTable),                   intent(in   ) :: table
    integer,                          optional, intent(  out) :: nEntries
    real, dimension(:),               optional, intent(  out) :: key, extinction, singleScatteringAlbedo
    character(len = *), dimension(:), optional, intent(  out) :: phaseFunctionDescriptions
    character(len = *),               optional, intent(  out) :: tableDescription
    type(ErrorMessage),                         intent(inout) :: status
    integer :: n

    if(.not. isReady_PhaseFunctionTable(table)) then
      call setStateToFailure(status, "getInfo_PhaseFunctionTable: table hasn't been initialized.")
      return
    end if
    n = size(table%phaseFunctions)
    if(present(nEntries)) nEntries = n
    if(present(key)) then
      if(size(key) < n) call setStateToFailure(status, "getInfo_PhaseFunctionTable: key array is too small.")
      if(size(key) >= n) key(:n) = table%key(:)
    end if
    if(present(extinction)) then
      if(size(extinction) < n) call setStateToFailure(status, "getInfo_PhaseFunctionTable: extinction array is too small.")
      if(size(extinction) >= n) extinction(:n) = table%phaseFunctions(:)%extinction
    end if
    if(present(singleScatteringAlbedo)) then
      if(size(singleScatteringAlbedo) < n) &
        call setStateToFailure(status, "getInfo_PhaseFunctionTable: singleScatteringAlbedo array is too small.")
      if(size(singleScatteringAlbedo) >= n) singleScatteringAlbedo(:n) = table%phaseFunctions(:)%singleScatteringAlbedo
    end if
    if(present(phaseFunctionDescriptions)) then
      if(size(phaseFunctionDescriptions) >= n) phaseFunctionDescriptions(:n) = table%phaseFunctions(:)%description
    end if
    if(present(tableDescription)) tableDescription = table%description
    if(.not. stateIsFailure(status)) call setStateToSuccess(status)
  end subroutine getInfo_PhaseFunctionTable

  ! Element n of a table (a deep copy: finalize it when done)
  function getElement(n, table, status)
    integer,                  intent(in   ) :: n
    type(phaseFunctionTable), intent(in   ) :: table
    type(ErrorMessage),       intent(inout) :: status
    type(phaseFunction)                     :: getElement
    if(.not. isReady_PhaseFunctionTable(table)) then
      call setStateToFailure(status, "getElement: phase function table hasn't been initialized.")
    else if(n < 1 .or. n > size(table%phaseFunctions)) then
      call setStateToFailure(status, "getElement: asking for non-existent element.")
    else
      getElement = copy_PhaseFunction(table%phaseFunctions(n))
      call setStateToSuccess(status)
    end if
  end function getElement

  ! ------------------------------------------------------------------------------------------------
  ! Evaluation
  ! ------------------------------------------------------------------------------------------------
  subroutine checkRequestedAngles(angles, nOut, status)
    real, dimension(:), intent(in   ) :: angles
    integer,            intent(in   ) :: nOut
    type(ErrorMessage), intent(inout) :: status
    if(any(angles < smallestAngle) .or. any(angles > largestAngle)) &
      call setStateToFailure(status, "getPhaseFunctionValues: ScatteringAngle out of bounds.")
    if(size(angles) /= nOut) &
      call setStateToFailure(status, "getPhaseFunctionValues: Number of scattering angles and phase function values must match.")
  end subroutine checkRequestedAngles

  ! linear interpolation in cos(angle) between the stored pairs bracketing each requested angle
  subroutine interpolateInCosine(stored, storedValues, angles, values, chainGuesses)
    real, dimension(:), intent(in ) :: stored, storedValues, angles
    real, dimension(:), intent(out) :: values
    logical,            intent(in ) :: chainGuesses
    integer :: i, lo, hi, nStored
    real    :: dMu, w
    nStored = size(stored)
    lo = 0
    do i = 1, size(angles)
      if(chainGuesses .and. i > 1) then
        lo = findIndex(angles(i), stored, firstGuess = lo)
      else
        lo = findIndex(angles(i), stored)
      end if
      if(lo < nStored) then
        hi = lo + 1
        dMu = cos(stored(hi)) - cos(stored(lo))
      else
        hi = lo                       ! requested angle is the last stored angle: weight falls on it
        dMu = huge(dMu)
      end if
      w = 1. - (cos(angles(i)) - cos(stored(lo))) / dMu
      values(i) = w * storedValues(lo) + (1. - w) * storedValues(hi)
    end do
  end subroutine interpolateInCosine

  subroutine valuesOfOne(phaseFunctionVar, scatteringAngle, value, status)
    type(phaseFunction), intent(in   ) :: phaseFunctionVar
    real, dimension(:),  intent(in   ) :: scatteringAngle
    real, dimension(:),  intent(  out) :: value
    type(ErrorMessage),  intent(inout) :: status
    integer :: l, maxL
    real, dimension(:, :), allocatable :: P

    if(.not. isReady_PhaseFunction(phaseFunctionVar)) &
      call setStateToFailure(status, "getPhaseFunctionValues: Phase function variable has not been initialized.")
    call checkRequestedAngles(scatteringAngle, size(value), status)
    if(stateIsFailure(status)) return
    if(associated(phaseFunctionVar%legendreCoefficients)) then
      maxL = size(phaseFunctionVar%legendreCoefficients)
      if(maxL == 0) then
        value(:) = 0.5                ! isotropic
      else
        allocate(P(0:maxL, size(scatteringAngle)))
        P(:, :) = computeLegendrePolynomials(maxL, cos(scatteringAngle))
        value(:) = matmul((/ 1., phaseFunctionVar%legendreCoefficients(:) /) * (/ (2 * l + 1, l = 0, maxL) /), P)
        deallocate(P)
      end if
    else
      call interpolateInCosine(phaseFunctionVar%scatteringAngle, phaseFunctionVar%value, scatteringAngle, value, .false.)
    end if
    call setStateToSuccess(status)
  end subroutine valuesOfOne

  ! values(nAngles, nEntries)
  subroutine valuesOfTable(table, scatteringAngle, values, status)
    type(phaseFunctionTable), intent(in   ) :: table
    real, dimension(:),       intent(in   ) :: scatteringAngle
    real, dimension(:, :),    intent(  out) :: values
    type(ErrorMessage),       intent(inout) :: status
    integer :: i, l, maxL, nA
    real, dimension(:, :), allocatable :: scaledP

    nA = size(scatteringAngle)
    if(.not. isReady_PhaseFunctionTable(table)) then
      call setStateToFailure(status, "getPhaseFunctionValues: Phase function table has not been initialized.")
      return
    end if
    call checkRequestedAngles(scatteringAngle, size(values, 1), status)
    if(any(scatteringAngle(2:) - scatteringAngle(:nA - 1) <= 0.)) &
      call setStateToFailure(status, "getPhaseFunctionValues: Scattering angle must be increasing, unique.")
    if(size(table%phaseFunctions) /= size(values, 2)) &
      call setStateToFailure(status, "getPhaseFunctionValues: Number of scattering angles and phase function values must match.")
    if(stateIsFailure(status)) return

    maxL = 0
    do i = 1, size(table%phaseFunctions)
      if(associated(table%phaseFunctions(i)%legendreCoefficients)) &
        maxL = max(maxL, size(table%phaseFunctions(i)%legendreCoefficients))
    end do
    if(maxL > 0) then     ! (2l+1) P_l once for all Legendre entries
      allocate(scaledP(0:maxL, nA))
      scaledP(:, :) = spread((/ (2 * l + 1, l = 0, maxL) /), dim = 2, ncopies = nA) * &
                      computeLegendrePolynomials(maxL, cos(scatteringAngle))
    end if
    do i = 1, size(table%phaseFunctions)
      if(associated(table%phaseFunctions(i)%legendreCoefficients)) then
        l = size(table%phaseFunctions(i)%legendreCoefficients)
        if(l == 0) then
          values(:, i) = 0.5
        else
          values(:, i) = matmul((/ 1., table%phaseFunctions(i)%legendreCoefficients(:) /), scaledP(0:l, :))
        end if
      else
        call interpolateInCosine(table%phaseFunctions(i)%scatteringAngle, table%phaseFunctions(i)%value, &
                                 scatteringAngle, values(:, i), .true.)
      end if
    end do
    if(allocated(scaledP)) deallocate(scaledP)
    call setStateToSuccess(status)
  end subroutine valuesOfTable

  ! Legendre coefficients l = 1.. of a phase function (stored ones, or by Lobatto quadrature of the pairs)
  subroutine getPhaseFunctionCoefficients(phaseFunctionVar, legendreCoefficients, status)
    type(phaseFunction), intent(in   ) :: phaseFunctionVar
    real, dimension(:),  intent(  out) :: legendreCoefficients
    type(ErrorMessage),  intent(inout) :: status
    integer :: nStored, nWanted, nQuad
    real, dimension(:),    allocatable :: mus, weights, vals
    real, dimension(:, :), allocatable :: P

    if(.not. isReady_PhaseFunction(phaseFunctionVar)) then
      call setStateToFailure(status, "getPhaseFunctionCoefficients: Phase function variable has not been initialized.")
      return
    end if
    nWanted = size(legendreCoefficients)
    if(associated(phaseFunctionVar%legendreCoefficients)) then
      nStored = size(phaseFunctionVar%legendreCoefficients)
      legendreCoefficients(:) = 0.
      legendreCoefficients(:min(nStored, nWanted)) = phaseFunctionVar%legendreCoefficients(:min(nStored, nWanted))
    else
      nQuad = 2 * size(phaseFunctionVar%scatteringAngle)
      allocate(mus(nQuad), weights(nQuad), vals(nQuad), P(0:nWanted, nQuad))
      call computeLobattoTerms(mus, weights)
      mus = mus(nQuad:1:-1)
      P(:, :) = computeLegendrePolynomials(nWanted, mus)
      call valuesOfOne(phaseFunctionVar, acos(mus), vals, status)
      legendreCoefficients(:) = 0.5 * matmul(P(1:, :), weights * vals)
      deallocate(mus, weights, vals, P)
    end if
    call setStateToSuccess(status)
  end subroutine getPhaseFunctionCoefficients

  ! ------------------------------------------------------------------------------------------------
  ! File I/O (netCDF classic): layout of Code/scatteringPhaseFunctions.f95:928-1252
  ! ------------------------------------------------------------------------------------------------
  include 'phaseFunctionTableIO.inc'
end module scatteringPhaseFunctions
