! Fortran-95 shell of the MI355X photon-tracing integrator -- surface description.
! Public interface of the reference's module surfaceProperties (Code/surfaceProperties.f95:52-56).  The
! surface model is the reference's example: a Lambertian albedo per rectangle of an x-y grid
! (numberOfParameters = 1); the GPU kernel evaluates it, so the shell also exposes the raw grid
! (getSurfaceGrid) for the device copy.
module surfaceProperties
  use ErrorMessages,    only: ErrorMessage, stateIsFailure, setStateToFailure, setStateToSuccess
  use numericUtilities, only: findIndex
  implicit none
  private
  integer, parameter :: numberOfParameters = 1

  type surfaceDescription
    private
    real, dimension(:),       pointer :: xPosition => null(), yPosition => null()
    real, dimension(:, :, :), pointer :: BRDFParameters => null()
  end type surfaceDescription

  interface new_SurfaceDescription
    module procedure surfaceOnGrid, surfaceUniform
  end interface

  public :: surfaceDescription
  public :: new_SurfaceDescription, copy_surfaceDescription, finalize_SurfaceDescription, &
            isReady_surfaceDescription, computeSurfaceReflectance
  public :: getSurfaceGrid   ! extension used by the GPU integrator
contains
  function surfaceOnGrid(surfaceParameters, xPosition, yPosition, status) result(thisSurfaceDescription)
    real, dimension(:, :, :), intent(in   ) :: surfaceParameters
    real, dimension(:),       intent(in   ) :: xPosition, yPosition
    type(ErrorMessage),       intent(inout) :: status
    type(surfaceDescription)                :: thisSurfaceDescription
    integer :: nx, ny

    nx = size(xPosition) - 1; ny = size(yPosition) - 1
    if(size(surfaceParameters, 1) /= numberOfParameters) &
      call setStateToFailure(status, "new_SurfaceDescription: Wrong number of parameters supplied for surface BRDF.")
    if(size(surfaceParameters, 2) /= nx .or. size(surfaceParameters, 3) /= ny) &
      call setStateToFailure(status, "new_SurfaceDescription: position vector(s) are incorrect length.")
    if(any(xPosition(2:) - xPosition(:nx) <= 0.) .or. any(yPosition(2:) - yPosition(:ny) <= 0.)) &
      call setStateToFailure(status, "new_SurfaceDescription: positions must be unique, increasing.")
    if(any(surfaceParameters(1, :, :) < 0.) .or. any(surfaceParameters(1, :, :) > 1.)) &
      call setStateToFailure(status, "new_SurfaceDescription: surface reflectance must be between 0 and 1")
    if(stateIsFailure(status)) return
    allocate(thisSurfaceDescription%xPosition(nx + 1), thisSurfaceDescription%yPosition(ny + 1), &
             thisSurfaceDescription%BRDFParameters(numberOfParameters, nx, ny))
    thisSurfaceDescription%xPosition(:) = xPosition(:)
    thisSurfaceDescription%yPosition(:) = yPosition(:)
    thisSurfaceDescription%BRDFParameters(:, :, :) = surfaceParameters(:, :, :)
    call setStateToSuccess(status)
  end function surfaceOnGrid

  ! horizontally uniform surface: one rectangle covering (0, huge) in both directions
  function surfaceUniform(surfaceParameters, status) result(thisSurfaceDescription)
    real, dimension(:), intent(in   ) :: surfaceParameters
    type(ErrorMessage), intent(inout) :: status
    type(surfaceDescription)          :: thisSurfaceDescription
    if(size(surfaceParameters) /= numberOfParameters) then
      call setStateToFailure(status, "new_SurfaceDescription: Wrong number of parameters supplied for surface BRDF.")
    else
      thisSurfaceDescription = surfaceOnGrid(reshape(surfaceParameters, (/ numberOfParameters, 1, 1 /)), &
                                             (/ 0., huge(1.) /), (/ 0., huge(1.) /), status)
    end if
  end function surfaceUniform

  pure function wrapInto(a, lower, upper) result(w)
    real, intent(in) :: a, lower, upper
    real             :: w
    w = a
    do while(.not. (w <= upper .and. w > lower))
      if(w > upper) then
        w = w - (upper - lower)
      else if(w == lower) then
        w = upper
      else
        w = w + (upper - lower)
      end if
    end do
  end function wrapInto

  pure function computeSurfaceReflectance(thisSurfaceDescription, xPos, yPos, &
                                          incomingMu, outgoingMu, incomingPhi, outgoingPhi) result(surfaceReflectance)
    type(surfaceDescription), intent(in) :: thisSurfaceDescription
    real,                     intent(in) :: xPos, yPos, incomingMu, outgoingMu, incomingPhi, outgoingPhi
    real                                 :: surfaceReflectance
    integer :: ix, iy, nx, ny
    nx = size(thisSurfaceDescription%xPosition); ny = size(thisSurfaceDescription%yPosition)
    ix = findIndex(wrapInto(xPos, thisSurfaceDescription%xPosition(1), thisSurfaceDescription%xPosition(nx)), &
                   thisSurfaceDescription%xPosition)
    iy = findIndex(wrapInto(yPos, thisSurfaceDescription%yPosition(1), thisSurfaceDescription%yPosition(ny)), &
                   thisSurfaceDescription%yPosition)
    surfaceReflectance = lambertian(thisSurfaceDescription%BRDFParameters(:, ix, iy))
  end function computeSurfaceReflectance

  pure function lambertian(parameters) result(R)
    real, dimension(numberOfParameters), intent(in) :: parameters
    real                                            :: R
    R = parameters(1)
  end function lambertian

  elemental function isReady_surfaceDescription(thisSurface)
    type(surfaceDescription), intent(in) :: thisSurface
    logical                              :: isReady_surfaceDescription
    isReady_surfaceDescription = associated(thisSurface%xPosition) .and. associated(thisSurface%yPosition) .and. &
                                 associated(thisSurface%BRDFParameters)
  end function isReady_surfaceDescription

  function copy_surfaceDescription(original) result(copy)
    type(surfaceDescription), intent(in) :: original
    type(surfaceDescription)             :: copy
    if(.not. isReady_surfaceDescription(original)) return
    allocate(copy%xPosition(size(original%xPosition)), copy%yPosition(size(original%yPosition)), &
             copy%BRDFParameters(size(original%BRDFParameters, 1), size(original%BRDFParameters, 2), &
                                 size(original%BRDFParameters, 3)))
    copy%xPosition(:) = original%xPosition(:)
    copy%yPosition(:) = original%yPosition(:)
    copy%BRDFParameters(:, :, :) = original%BRDFParameters(:, :, :)
  end function copy_surfaceDescription

  subroutine finalize_surfaceDescription(thisSurface)
    type(surfaceDescription), intent(inout) :: thisSurface
    if(associated(thisSurface%xPosition))      deallocate(thisSurface%xPosition)
    if(associated(thisSurface%yPosition))      deallocate(thisSurface%yPosition)
    if(associated(thisSurface%BRDFParameters)) deallocate(thisSurface%BRDFParameters)
  end subroutine finalize_surfaceDescription

  ! edges and albedo(nx, ny) for the device copy
  subroutine getSurfaceGrid(thisSurface, xEdges, yEdges, albedo)
    type(surfaceDescription), intent(in) :: thisSurface
    real, dimension(:),    pointer :: xEdges, yEdges
    real, dimension(:, :), pointer :: albedo
    xEdges => thisSurface%xPosition
    yEdges => thisSurface%yPosition
    albedo => thisSurface%BRDFParameters(1, :, :)
  end subroutine getSurfaceGrid
end module surfaceProperties
