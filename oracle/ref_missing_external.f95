! oracle/ref_missing_external.f95 -- test infrastructure; this repository's own code, linked into oracle/_ref/ref_loop and ref_streams.
!
! The reference does not link as it is shipped: Code/monteCarloIllumination.f95:17-19 imports ErrorMessages with an only-list that
! lacks setStateToWarning and calls it all the same (:266, :268, :374-379: the flux / intensity / spotlight streams' checks), which
! leaves an EXTERNAL reference `setstatetowarning_` in its object.  This is that external -- it hands on to the module procedure it
! was meant to be; a Directional stream, the only kind this program makes, never gets there.
subroutine setStateToWarning(messageVariable, messageText)
  use ErrorMessages, only: ErrorMessage, moduleProcedure => setStateToWarning
  implicit none
  type(ErrorMessage), intent(inout) :: messageVariable
  character(len = *), intent(in)    :: messageText
  call moduleProcedure(messageVariable, messageText)
end subroutine setStateToWarning
