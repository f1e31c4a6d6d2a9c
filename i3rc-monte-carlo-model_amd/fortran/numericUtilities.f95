! Fortran-95 shell of the MI355X photon-tracing integrator -- numerical helpers (host side, setup time).
! Public interface of the reference's module numericUtilities (Code/numericUtilities.f95:11-12):
! findIndex, computeLobattoTerms, computeGaussLegendreTerms, computeLegendrePolynomials.
module numericUtilities
  implicit none
  private
  public :: computeLobattoTerms, computeGaussLegendreTerms, computeLegendrePolynomials, findIndex
contains
  ! Values of P_0 .. P_maxL at each mu (three-term recurrence), result(0:maxL, size(mus))
  pure function computeLegendrePolynomials(maxL, mus) result(legendreP)
    integer,            intent(in) :: maxL
    real, dimension(:), intent(in) :: mus
    real, dimension(0:maxL, size(mus)) :: legendreP
    integer :: l
    legendreP(0, :) = 1.
    if(maxL >= 1) legendreP(1, :) = mus(:)
    do l = 1, maxL - 1
      legendreP(l + 1, :) = ((2 * l + 1) * mus(:) * legendreP(l, :) - l * legendreP(l - 1, :)) / (l + 1)
    end do
  end function computeLegendrePolynomials

  ! Largest i with table(i) <= value (table increasing), optionally hunting outwards from firstGuess first.
  ! Returns 0 below the table and size(table) at or above its last entry.
  pure function findIndex(value, table, firstGuess)
    real,               intent(in) :: value
    real, dimension(:), intent(in) :: table
    integer, optional,  intent(in) :: firstGuess
    integer                        :: findIndex
    integer :: lo, hi, mid, stride, n

    n = size(table)
    if(present(firstGuess)) then
      lo = firstGuess
      stride = 1
      do
        hi = min(lo + stride, n)
        if(lo == n) exit
        if(table(lo) <= value .and. table(hi) > value) exit
        if(table(lo) > value) then
          hi = lo
          lo = max(hi - stride, 1)
        else
          lo = hi
        end if
        stride = 2 * stride
      end do
    else
      lo = 0
      hi = n
    end if
    do while(lo /= n .and. hi > lo + 1)
      mid = (lo + hi) / 2
      if(value >= table(mid)) then
        lo = mid
      else
        hi = mid
      end if
    end do
    findIndex = lo
  end function findIndex

  ! n-point Gauss-Lobatto abscissas (end points included) and weights on [-1, 1]: the interior nodes are
  ! the zeros of P'_{n-1}, found by Newton iteration from a trigonometric first guess.
  pure subroutine computeLobattoTerms(mus, weights)
    real, dimension(:), intent(out) :: mus, weights
    integer, parameter :: newtonLimit = 25
    real,    parameter :: tolerance = 3.
    integer :: n, half, nRoots, k, sweep
    real    :: pi, offset
    real, dimension(:),    allocatable :: root, previous, d1, d2
    real, dimension(:, :), allocatable :: P
    logical, dimension(:), allocatable :: moving

    n = min(size(mus), size(weights))
    pi = acos(-1.)
    half = (n + 1) / 2
    nRoots = half - 1
    allocate(root(nRoots), previous(nRoots), d1(nRoots), d2(nRoots), moving(nRoots), P(0:n - 1, nRoots))
    offset = 0.5
    if(mod(n, 2) == 1) offset = 1.
    root(:) = sin(pi * ((/ (real(k), k = 1, nRoots) /) - offset) / (n - 1. + .5))

    moving(:) = .true.
    sweep = 0
    do
      P(:, :) = computeLegendrePolynomials(n - 1, root)
      where(moving)
        d1 = (n - 1) * (root * P(n - 1, :) - P(n - 2, :)) / (root**2 - 1.)
        d2 = (2. * root * d1 - (n * (n - 1) * P(n - 1, :))) / (1. - root**2)
        previous = root
        root = root - d1 / d2
      end where
      moving(:) = abs(root - previous) > tolerance * spacing(root)
      if(.not. any(moving)) exit
      sweep = sweep + 1
      if(sweep > newtonLimit + 1) exit
    end do

    mus(:) = 0.; weights(:) = 0.
    mus(1) = -1.
    weights(1) = 2. / (n * (n - 1))
    do k = 1, nRoots
      mus(half + 1 - k)     = -root(k)
      weights(half + 1 - k) = 2. / (n * (n - 1) * P(n - 1, k)**2)
    end do
    do k = 1, n / 2            ! mirror the negative half onto the positive one
      mus(n + 1 - k)     = -mus(k)
      weights(n + 1 - k) = weights(k)
    end do
    if(mod(n, 2) == 1) mus(half) = 0.
    deallocate(root, previous, d1, d2, moving, P)
  end subroutine computeLobattoTerms

  ! n-point Gauss-Legendre abscissas and weights on (-1, 1): zeros of P_n by Newton iteration, the same scheme as the
  ! Lobatto nodes above (a node stops moving once a step is within 2 spacing() of it; the weight uses the derivative
  ! of the node's last step).  Held bit for bit against the reference's own routine: tests/test_ref_numerics.py.
  pure subroutine computeGaussLegendreTerms(mus, weights)
    real, dimension(:), intent(out) :: mus, weights
    integer, parameter :: newtonLimit = 25
    real,    parameter :: tolerance = 2.
    integer :: n, half, k, sweep
    real    :: pi
    real, dimension(:),    allocatable :: root, previous, deriv
    real, dimension(:, :), allocatable :: P
    logical, dimension(:), allocatable :: moving

    n = min(size(mus), size(weights))
    pi = acos(-1.)
    half = (n + 1) / 2
    allocate(root(half), previous(half), deriv(half), moving(half), P(0:n, half))
    root(:) = cos(pi * ((/ (real(k), k = 1, half) /) - .25) / (n + .5))
    moving(:) = .true.
    sweep = 0
    do
      P(:, :) = computeLegendrePolynomials(n, root)
      where(moving)
        deriv = n * (root * P(n, :) - P(n - 1, :)) / (root**2 - 1.)
        previous = root
        root = root - P(n, :) / deriv
      end where
      moving(:) = abs(root - previous) > tolerance * spacing(root)
      if(.not. any(moving)) exit
      sweep = sweep + 1
      if(sweep > newtonLimit + 1) exit
    end do

    mus(:) = 0.; weights(:) = 0.
    do k = 1, half
      mus(k)     = -root(k)
      weights(k) = 2. / ((1. - root(k)**2) * deriv(k)**2)
    end do
    do k = 1, n / 2            ! mirror the negative half onto the positive one
      mus(n + 1 - k)     = -mus(k)
      weights(n + 1 - k) = weights(k)
    end do
    if(mod(n, 2) == 1) mus(half) = -mus(half)   ! (the middle node mirrors onto itself: its sign flips, as the reference's does)
    deallocate(root, previous, deriv, moving, P)
  end subroutine computeGaussLegendreTerms
end module numericUtilities
