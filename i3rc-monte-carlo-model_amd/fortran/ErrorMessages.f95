! Fortran-95 shell of the MI355X photon-tracing integrator -- status object.
! Keeps the public interface and observable behaviour of the reference's module ErrorMessages
! (Code/ErrorMessages.f95:31-71): a bounded history of (state, text) records; the newest record is "current".
module ErrorMessages
  implicit none
  private

  integer, parameter :: historyDepth = 100, textLength = 256
  integer, parameter :: isUndefined = 0, isSuccess = 1, isWarning = 2, isFailure = 3
  character(len = *), parameter :: undefinedText = "Status is undefined."

  type, public :: ErrorMessage
    private
    integer :: newest = 0, cursor = 0
    integer,                    dimension(0:historyDepth) :: state = isUndefined
    character(len = textLength), dimension(0:historyDepth) :: text  = undefinedText
  end type ErrorMessage

  public :: stateIsSuccess, stateIsWarning, stateIsFailure
  public :: initializeState, setStateToSuccess, setStateToWarning, setStateToFailure, setStateToCompleteSuccess
  public :: firstMessage, nextMessage, getCurrentMessage, moreMessagesExist
  public :: getErrorMessageLimits
contains
  ! -- one place where records are appended ------------------------------------------------------
  subroutine appendRecord(m, newState, messageText)
    type(ErrorMessage),           intent(inout) :: m
    integer,                      intent(in   ) :: newState
    character(len = *), optional, intent(in   ) :: messageText
    m%newest = min(m%newest + 1, historyDepth)   ! a full history overwrites its last slot
    m%cursor = m%newest
    m%state(m%newest) = newState
    m%text (m%newest) = ""
    if(present(messageText)) m%text(m%newest) = trim(messageText)
  end subroutine appendRecord

  logical function currentStateIs(m, what)
    type(ErrorMessage), intent(in) :: m
    integer,            intent(in) :: what
    currentStateIs = m%newest > 0 .and. m%state(m%cursor) == what
  end function currentStateIs

  logical function stateIsSuccess(messageVariable)
    type(ErrorMessage), intent(in) :: messageVariable
    stateIsSuccess = currentStateIs(messageVariable, isSuccess)
  end function stateIsSuccess

  logical function stateIsWarning(messageVariable)
    type(ErrorMessage), intent(in) :: messageVariable
    stateIsWarning = currentStateIs(messageVariable, isWarning)
  end function stateIsWarning

  logical function stateIsFailure(messageVariable)
    type(ErrorMessage), intent(in) :: messageVariable
    stateIsFailure = currentStateIs(messageVariable, isFailure)
  end function stateIsFailure

  subroutine initializeState(messageVariable)
    type(ErrorMessage), intent(out) :: messageVariable
    messageVariable%newest = 0
    messageVariable%cursor = 0
    messageVariable%state(:) = isUndefined
    messageVariable%text(:)  = ""
    messageVariable%text(0)  = undefinedText
  end subroutine initializeState

  subroutine setStateToSuccess(messageVariable, messageText)
    type(ErrorMessage),           intent(inout) :: messageVariable
    character(len = *), optional, intent(in   ) :: messageText
    call appendRecord(messageVariable, isSuccess, messageText)
  end subroutine setStateToSuccess

  subroutine setStateToWarning(messageVariable, messageText)
    type(ErrorMessage),           intent(inout) :: messageVariable
    character(len = *), optional, intent(in   ) :: messageText
    call appendRecord(messageVariable, isWarning, messageText)
  end subroutine setStateToWarning

  subroutine setStateToFailure(messageVariable, messageText)
    type(ErrorMessage),           intent(inout) :: messageVariable
    character(len = *), optional, intent(in   ) :: messageText
    call appendRecord(messageVariable, isFailure, messageText)
  end subroutine setStateToFailure

  ! Clears the history, then records one success.
  subroutine setStateToCompleteSuccess(messageVariable, messageText)
    type(ErrorMessage),           intent(out) :: messageVariable
    character(len = *), optional, intent(in ) :: messageText
    call initializeState(messageVariable)
    call appendRecord(messageVariable, isSuccess, messageText)
  end subroutine setStateToCompleteSuccess

  ! -- iteration over the history -----------------------------------------------------------------
  subroutine firstMessage(messageVariable)
    type(ErrorMessage), intent(inout) :: messageVariable
    messageVariable%cursor = min(1, messageVariable%newest)
  end subroutine firstMessage

  subroutine nextMessage(messageVariable)
    type(ErrorMessage), intent(inout) :: messageVariable
    messageVariable%cursor = messageVariable%cursor + 1
  end subroutine nextMessage

  logical function moreMessagesExist(messageVariable)
    type(ErrorMessage), intent(in) :: messageVariable
    moreMessagesExist = messageVariable%cursor <= messageVariable%newest
  end function moreMessagesExist

  function getCurrentMessage(messageVariable)
    type(ErrorMessage), intent(in) :: messageVariable
    character(len = textLength)    :: getCurrentMessage
    if(messageVariable%cursor <= 0 .or. messageVariable%cursor > historyDepth) then
      getCurrentMessage = undefinedText
    else
      getCurrentMessage = messageVariable%text(messageVariable%cursor)
    end if
  end function getCurrentMessage

  subroutine getErrorMessageLimits(messageVariable, maxNumberOfMessages, maxMessageLength)
    type(ErrorMessage), intent(in ) :: messageVariable
    integer, optional,  intent(out) :: maxNumberOfMessages, maxMessageLength
    if(present(maxNumberOfMessages)) maxNumberOfMessages = historyDepth
    if(present(maxMessageLength))    maxMessageLength    = textLength
  end subroutine getErrorMessageLimits
end module ErrorMessages
