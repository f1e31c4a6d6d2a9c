! oracle/ref_dump.f95 -- TEST INFRASTRUCTURE.  A caller, written for this repository, of three modules' public
! procedures: numericUtilities (findIndex, computeLobattoTerms, computeGaussLegendreTerms, computeLegendrePolynomials)
! and surfaceProperties (new_SurfaceDescription, computeSurfaceReflectance), with ErrorMessages underneath.
!
! Linked against the REFERENCE's modules -- compiled unmodified and in place from /root/reference/Code/
! {ErrorMessages,numericUtilities,surfaceProperties}.f95, the three modules of the path that need no netCDF -- it is
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
program refDump
  use ErrorMessages
  use numericUtilities
  use surfaceProperties
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
end program refDump
