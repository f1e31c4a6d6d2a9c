! Fortran-95 shell of the MI355X photon-tracing integrator -- inverse (cumulative) phase-function tables.
! Public interface of the reference's module inversePhaseFunctions (Code/inversePhaseFunctions.f95:24-25):
! computeInversePhaseFuncTable(forwardTable, inverseTable(nSteps, nEntries), status).
! Column i of the result is the scattering angle (radians) at nSteps equally spaced values of the cumulative
! probability, for entry i of the table; it is what the GPU kernel samples scattering angles from.
module inversePhaseFunctions
  use ErrorMessages,            only: ErrorMessage, stateIsFailure, setStateToFailure, setStateToSuccess
  use scatteringPhaseFunctions, only: phaseFunction, phaseFunctionTable, isReady_PhaseFunctionTable, &
                                      getInfo_PhaseFunction, getInfo_PhaseFunctionTable, getElement, &
                                      getPhaseFunctionValues, finalize_PhaseFunction
  use numericUtilities,         only: findIndex, computeLobattoTerms
  implicit none
  private
  public :: computeInversePhaseFuncTable
contains
  subroutine computeInversePhaseFuncTable(forwardTable, inverseTable, status)
    type(phaseFunctionTable), intent(in   ) :: forwardTable
    real, dimension(:, :),    intent(  out) :: inverseTable
    type(ErrorMessage),       intent(inout) :: status
    integer             :: nEntries, i
    type(phaseFunction) :: one
    type(ErrorMessage)  :: scratch

    if(.not. isReady_PhaseFunctionTable(forwardTable)) then
      call setStateToFailure(status, "computeInversePhaseFunctionTable: Forward table isn't ready.")
      return
    end if
    call getInfo_PhaseFunctionTable(forwardTable, nEntries = nEntries, status = status)
    if(size(inverseTable, 2) /= nEntries) call setStateToFailure(status, &
      "computeInversePhaseFunctionTable: Array for inverse table has the wrong number of entries")
    if(stateIsFailure(status)) return
    do i = 1, nEntries
      one = getElement(i, forwardTable, scratch)
      call invertOne(one, inverseTable(:, i), status)
      call finalize_PhaseFunction(one)
      if(stateIsFailure(status)) then
        call setStateToFailure(status, "computeInversePhaseFunctionTable: Can't compute inverse tables.")
        return
      end if
    end do
    call setStateToSuccess(status)
  end subroutine computeInversePhaseFuncTable

  ! Cumulative distribution in mu = cos(angle) by the trapezoid rule on the native angles (angle/value pairs)
  ! or on max(nMoments, 2) Lobatto nodes (Legendre expansion); then, for each probability step, the mu at
  ! which a piecewise-linear phase function reaches that cumulative value (closed form per interval).
  subroutine invertOne(p, angleOfProbability, status)
    type(phaseFunction), intent(in   ) :: p
    real, dimension(:),  intent(  out) :: angleOfProbability
    type(ErrorMessage),  intent(inout) :: status
    integer :: nSteps, nNodes, nMoments, nAngles, i, k
    real    :: prob, dCdf, radicand
    real, dimension(:), allocatable :: mu, f, cdf, scratch

    nSteps = size(angleOfProbability)
    call getInfo_PhaseFunction(p, nCoefficients = nMoments, nAngles = nAngles, status = status)
    if(stateIsFailure(status)) return
    if(nAngles > 0) then
      nNodes = nAngles
      allocate(mu(nNodes), f(nNodes), cdf(nNodes), scratch(nNodes))
      call getInfo_PhaseFunction(p, nativeAngles = scratch, status = status)
      if(.not. stateIsFailure(status)) call getPhaseFunctionValues(p, scratch, f, status)
      mu(:) = cos(scratch(nNodes:1:-1))
      f(:)  = f(nNodes:1:-1)
    else
      nNodes = max(nMoments, 2)
      allocate(mu(nNodes), f(nNodes), cdf(nNodes), scratch(nNodes))
      call computeLobattoTerms(mus = mu, weights = scratch)
      call getPhaseFunctionValues(p, acos(mu(nNodes:1:-1)), f, status)
      f(:) = f(nNodes:1:-1)
    end if
    if(stateIsFailure(status)) then
      deallocate(mu, f, cdf, scratch)
      return
    end if

    cdf(1) = 0.
    do i = 2, nNodes
      cdf(i) = cdf(i - 1) + (mu(i) - mu(i - 1)) * 0.5 * (f(i) + f(i - 1))
    end do
    cdf(:) = cdf(:) / cdf(nNodes)

    k = findIndex(0., cdf)
    do i = 1, nSteps - 1
      prob = float(i - 1) / float(nSteps - 1)
      if(i > 1) k = findIndex(prob, cdf, firstGuess = k)
      dCdf = cdf(k + 1) - cdf(k)
      if(dCdf <= spacing(cdf(k))) then                               ! no probability in this interval
        angleOfProbability(i) = acos(mu(k))
      else if(abs(f(k) - f(k + 1)) <= spacing(f(k))) then            ! locally constant phase function
        angleOfProbability(i) = acos(mu(k) + (mu(k + 1) - mu(k)) * (prob - cdf(k)) / dCdf)
      else                                                           ! locally linear phase function
        radicand = ((cdf(k + 1) - prob) * f(k)**2 + (prob - cdf(k)) * f(k + 1)**2) / dCdf
        angleOfProbability(i) = acos(mu(k) + (mu(k + 1) - mu(k)) / (f(k) - f(k + 1)) * (f(k) - sqrt(radicand)))
      end if
    end do
    angleOfProbability(nSteps) = 0.
    deallocate(mu, f, cdf, scratch)
    call setStateToSuccess(status)
  end subroutine invertOne
end module inversePhaseFunctions
