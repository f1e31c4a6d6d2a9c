! Fortran-95 shell of the MI355X photon-tracing integrator -- driver glue.
! Public interface of the reference's module UserInterface (Code/userInterface_Unix.f95:17): printStatus
! (prints warnings / failures and the message history, stops on failure) and getOneArgument.
module UserInterface
  use ErrorMessages,     only: ErrorMessage, stateIsSuccess, stateIsWarning, stateIsFailure, &
                               firstMessage, nextMessage, getCurrentMessage, moreMessagesExist
  use MultipleProcesses, only: MasterProc
  implicit none
  private
  public :: printStatus, getOneArgument
contains
  subroutine printStatus(status)
    type(ErrorMessage), intent(inout) :: status
    logical :: failed
    character(len = 256) :: line

    failed = stateIsFailure(status)
    if(failed .or. stateIsWarning(status)) then
      line = getCurrentMessage(status)
      if(len_trim(line) > 0) then
        print *, trim(line)
      else if(failed) then
        print *, "Status is Failure"
      else
        print *, "Status is warning"
      end if
    end if
    if(.not. stateIsSuccess(status)) then
      print *, "History:"
      call firstMessage(status)
      do while(moreMessagesExist(status))
        line = getCurrentMessage(status)
        if(len_trim(line) > 0) print *, "  ", trim(line)
        call nextMessage(status)
      end do
    end if
    if(failed) stop
  end subroutine printStatus

  function getOneArgument(message)
    character(len = *), optional, intent(in) :: message
    character(len = 256)                     :: getOneArgument
    if(present(message)) print *, message
    if(command_argument_count() < 1) then
      if(MasterProc) print *, "No file name supplied."
      stop
    end if
    call get_command_argument(1, getOneArgument)
    if(MasterProc) print *, 'Using value ' // trim(getOneArgument)
  end function getOneArgument
end module UserInterface
