! oracle/ref_dump.f95 -- TEST INFRASTRUCTURE.  A caller, written for this repository, of four modules' public
! procedures: numericUtilities (findIndex, computeLobattoTerms, computeGaussLegendreTerms, computeLegendrePolynomials),
! surfaceProperties (new_SurfaceDescription, computeSurfaceReflectance), ErrorMessages (the status object every procedure
! of the boundary reports through) and CharacterUtils.
!
! Linked against the REFERENCE's modules -- compiled unmodified and in place from /root/reference/Code/
! {ErrorMessages,numericUtilities,surfaceProperties,characterUtils}.f95, the modules of the path and its boundary that need no netCDF -- it is
! oracle/_ref/ref_dump (oracle/Makefile, target _ref; build container only) and writes what tests/golden/ref_numerics.npz
! holds (tests/golden/make_ref_numerics.py).  Linked against the shell's modules of the same names it is
! fortran/build/shellNumericsDump, whose output the tests hold against that fixture bit for bit.
!
! Protocol (stdin -> stdout, list-directed; every real travels as the default integer with the same 32 bits, so that
! nothing is rounded by formatted I/O):
!   findIndex n m g      table(n), values(m), [firstGuess(m) when g = 1]         -> m integers
!   lobatto n                                                                     -> mus(n), weights(n)
!   gauss n                                                                       -> mus(n), weights(n)
!   legendre maxL m      mus(m)                                                   -> P(0:maxL, m), l fastest
!   surface nx ny m      xPosition(nx+1), yPosition(ny+1), R(nx, ny) x fastest, x(m), y(m)  -> status, reflectance(m)
!   uniform m            R, x(m), y(m)                                             -> status, reflectance(m)
!   errors k             k lines "X text": X = S / W / F (setStateTo... with the text), s / w / f (without a text), C / c
!                        (setStateToCompleteSuccess), I (initializeState)      -> after each: "isSuccess isWarning isFailure n" and the
!                        n messages of the history (firstMessage ... moreMessagesExist), in brackets; the limits; "end"
!   chars k              k lines "I integer" / "C string" / "R string"          -> [IntToChar] and its len / CharToInt / CharToReal's bits; "end"
program refDump
  use ErrorMessages
  use numericUtilities
  use surfaceProperties
  use CharacterUtils
  implicit none
  character(len = 32)  :: command
  character(len = 256) :: line
  integer :: ios

  do
    read(*, '(a)', iostat = ios) line
    if(ios /= 0) exit
    if(len_trim(line) == 0) cycle
    read(line, *) command
    select case(trim(command))
      case("findIndex"); call dumpFindIndex
      case("lobatto");   call dumpQuadrature(.true.)
      case("gauss");     call dumpQuadrature(.false.)
      case("legendre");  call dumpLegendre
      case("surface");   call dumpSurface(.false.)
      case("uniform");   call dumpSurface(.true.)
      case("errors");    call dumpErrors
      case("chars");     call dumpChars
      case default
        write(*, '(a)') "unknown " // trim(command)
        stop 2
    end select
  end do
contains
  subroutine readReals(a)
    real, dimension(:), intent(out) :: a
    integer, dimension(size(a)) :: bits
    read(*, *) bits
    a = transfer(bits, a)
  end subroutine readReals

  subroutine writeReals(a)
    real, dimension(:), intent(in) :: a
    integer :: i
    do i = 1, size(a)
      write(*, '(i0)') transfer(a(i), 1)
    end do
  end subroutine writeReals

  subroutine dumpFindIndex
    integer :: n, m, g, i
    real,    dimension(:), allocatable :: table, values
    integer, dimension(:), allocatable :: guess
    read(line, *) command, n, m, g
    allocate(table(n), values(m), guess(m))
    call readReals(table); call readReals(values)
    if(g == 1) read(*, *) guess
    write(*, '(a, 1x, i0)') "findIndex", m
    do i = 1, m
      if(g == 1) then
        write(*, '(i0)') findIndex(values(i), table, guess(i))
      else
        write(*, '(i0)') findIndex(values(i), table)
      end if
    end do
    deallocate(table, values, guess)
  end subroutine dumpFindIndex

  subroutine dumpQuadrature(lobatto)
    logical, intent(in) :: lobatto
    integer :: n
    real, dimension(:), allocatable :: mus, weights
    read(line, *) command, n
    allocate(mus(n), weights(n))
    if(lobatto) then
      call computeLobattoTerms(mus, weights)
    else
      call computeGaussLegendreTerms(mus, weights)
    end if
    write(*, '(a, 1x, i0)') trim(command), 2 * n
    call writeReals(mus); call writeReals(weights)
    deallocate(mus, weights)
  end subroutine dumpQuadrature

  subroutine dumpLegendre
    integer :: maxL, m, j
    real, dimension(:),    allocatable :: mus
    real, dimension(:, :), allocatable :: P
    read(line, *) command, maxL, m
    allocate(mus(m), P(0:maxL, m))
    call readReals(mus)
    P(:, :) = computeLegendrePolynomials(maxL, mus)
    write(*, '(a, 1x, i0)') "legendre", (maxL + 1) * m
    do j = 1, m
      call writeReals(P(:, j))
    end do
    deallocate(mus, P)
  end subroutine dumpLegendre

  subroutine dumpSurface(uniform)
    logical, intent(in) :: uniform
    integer :: nx, ny, m, i
    real, dimension(:),       allocatable :: xPos, yPos, x, y, flat, refl
    real, dimension(:, :, :), allocatable :: params
    type(surfaceDescription) :: surface
    type(ErrorMessage)       :: status
    if(uniform) then
      read(line, *) command, m
      nx = 1; ny = 1
    else
      read(line, *) command, nx, ny, m
    end if
    allocate(xPos(nx + 1), yPos(ny + 1), flat(nx * ny), params(1, nx, ny), x(m), y(m), refl(m))
    if(.not. uniform) then
      call readReals(xPos); call readReals(yPos)
    end if
    call readReals(flat)
    params(1, :, :) = reshape(flat, (/ nx, ny /))
    call readReals(x); call readReals(y)
    call initializeState(status)
    if(uniform) then
      surface = new_SurfaceDescription(flat(1:1), status)
    else
      surface = new_SurfaceDescription(params, xPos, yPos, status)
    end if
    write(*, '(a, 1x, i0)') trim(command), m + 1
    if(stateIsFailure(status)) then
      write(*, '(i0)') 1
      refl(:) = -1.
    else
      write(*, '(i0)') 0
      do i = 1, m   ! (the four angles do not enter a Lambertian reflectance: Code/surfaceProperties.f95:154-162)
        refl(i) = computeSurfaceReflectance(surface, x(i), y(i), 0.5, 0.5, 0., 0.)
      end do
      call finalize_SurfaceDescription(surface)
    end if
    call writeReals(refl)
    deallocate(xPos, yPos, flat, params, x, y, refl)
  end subroutine dumpSurface

  subroutine dumpErrors
    integer :: k, i, n, maxN, maxLen
    character(len = 400) :: op
    type(ErrorMessage)   :: status
    read(line, *) command, k
    write(*, '(a)') "errors"
    do i = 1, k
      read(*, '(a)') op
      select case(op(1:1))
        case("S"); call setStateToSuccess(status, trim(op(3:)))
        case("W"); call setStateToWarning(status, trim(op(3:)))
        case("F"); call setStateToFailure(status, trim(op(3:)))
        case("C"); call setStateToCompleteSuccess(status, trim(op(3:)))
        case("s"); call setStateToSuccess(status)
        case("w"); call setStateToWarning(status)
        case("f"); call setStateToFailure(status)
        case("c"); call setStateToCompleteSuccess(status)
        case("I"); call initializeState(status)
      end select
      write(*, '(3(i0, 1x))', advance = "no") merge(1, 0, stateIsSuccess(status)), merge(1, 0, stateIsWarning(status)), merge(1, 0, stateIsFailure(status))
      ! the history, as a driver prints it (UserInterface's printStatus: firstMessage ... moreMessagesExist)
      n = 0
      call firstMessage(status)
      do
        if(.not. moreMessagesExist(status)) exit
        n = n + 1
        call nextMessage(status)
      end do
      write(*, '(i0)') n
      call firstMessage(status)
      do
        if(.not. moreMessagesExist(status)) exit
        write(*, '(a)') "[" // trim(getCurrentMessage(status)) // "]"
        call nextMessage(status)
      end do
    end do
    call getErrorMessageLimits(status, maxNumberOfMessages = maxN, maxMessageLength = maxLen)
    write(*, '(a, 1x, i0, 1x, i0)') "limits", maxN, maxLen
    write(*, '(a)') "end"
  end subroutine dumpErrors

  subroutine dumpChars
    integer :: k, i, v
    character(len = 200) :: op
    read(line, *) command, k
    write(*, '(a)') "chars"
    do i = 1, k
      read(*, '(a)') op
      select case(op(1:1))
        case("I")
          read(op(3:), *) v
          write(*, '(a, 1x, i0)') "[" // trim(IntToChar(v)) // "]", len(IntToChar(v))
        case("C"); write(*, '(i0)') CharToInt(trim(op(3:)))
        case("R"); write(*, '(i0)') transfer(CharToReal(trim(op(3:))), 1)
      end select
    end do
    write(*, '(a)') "end"
  end subroutine dumpChars
end program refDump
