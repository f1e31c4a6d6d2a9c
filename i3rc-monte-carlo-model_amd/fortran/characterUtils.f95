! Fortran-95 shell of the MI355X photon-tracing integrator -- string/number conversions.
! Same public names as the reference's module CharacterUtils (Code/characterUtils.f95:14).
module CharacterUtils
  implicit none
  private
  integer, parameter :: fieldWidth = 25   ! (the length of the reference's results, Code/characterUtils.f95:13: tests/test_ref_numerics.py)
  public :: CharToInt, IntToChar, CharToReal
contains
  elemental function CharToInt(inputString)
    character(len = *), intent(in) :: inputString
    integer                        :: CharToInt
    integer :: ios
    read(inputString, *, iostat = ios) CharToInt
    if(ios /= 0) CharToInt = 0
  end function CharToInt

  elemental function IntToChar(inputInteger)
    integer, intent(in)       :: inputInteger
    character(len = fieldWidth) :: IntToChar
    write(IntToChar, '(I0)') inputInteger
  end function IntToChar

  elemental function CharToReal(inputString)
    character(len = *), intent(in) :: inputString
    real                           :: CharToReal
    integer :: ios
    read(inputString, *, iostat = ios) CharToReal
    if(ios /= 0) CharToReal = 0.
  end function CharToReal
end module CharacterUtils
