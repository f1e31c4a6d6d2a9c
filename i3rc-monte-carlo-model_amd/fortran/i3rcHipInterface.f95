! iso_c_binding view of include/i3rc_hip.h: the thin shim between the Fortran-95 shell and the HIP library.
module i3rcHipInterface
  use, intrinsic :: iso_c_binding
  implicit none
  public

  type, bind(C) :: i3rc_params
    real(c_float)      :: surfaceAlbedo = 0.
    integer(c_int32_t) :: useSurfaceBDRF = 0
    integer(c_int32_t) :: useRayTracing = 1
    integer(c_int32_t) :: useRussianRoulette = 1
    integer(c_int32_t) :: useHybridPhaseFunsForIntenCalcs = 0
    integer(c_int32_t) :: numOrdersOrigPhaseFunIntenCalcs = 0
    integer(c_int32_t) :: useRussianRouletteForIntensity = 0
    real(c_float)      :: zetaMin = 0.3
    integer(c_int32_t) :: limitIntensityContributions = 0
    real(c_float)      :: maxIntensityContribution = huge(1.)
  end type i3rc_params

  type, bind(C) :: i3rc_source
    integer(c_int32_t) :: kind = 0
    real(c_float)      :: solarMu = 0., solarAzimuth = 0.
    type(c_ptr)        :: x = c_null_ptr, y = c_null_ptr, z = c_null_ptr, mu = c_null_ptr, phi = c_null_ptr
  end type i3rc_source

  type, bind(C) :: i3rc_tally_layout
    integer(c_int64_t) :: fluxUp, fluxDown, fluxAbsorbed, volumeAbsorption, intensityByComponent, intensityExcess, &
                          counters, total
  end type i3rc_tally_layout

  type, bind(C) :: i3rc_moments_layout
    integer(c_int64_t) :: fluxUp, fluxDown, fluxAbsorbed, volumeAbsorption, intensity, absorbedProfile, &
                          meanFluxUp, meanFluxDown, meanFluxAbsorbed, meanIntensity, total
  end type i3rc_moments_layout

  integer, parameter :: I3RC_CNT_PHOTONS = 0, I3RC_CNT_DROPPED = 1, I3RC_NUM_COUNTERS = 16
  integer, parameter :: I3RC_MAX_COMPONENTS = 255, I3RC_MAX_DIRECTIONS = 255

  interface
    function i3rc_hip_create(h, device, nx, ny, nz, ncomp, xEdges, yEdges, zEdges, totalExt, cumExt, ssa, pfIndex) &
             bind(C, name = "i3rc_hip_create") result(rc)
      import
      type(c_ptr),           intent(out) :: h
      integer(c_int), value              :: device, nx, ny, nz, ncomp
      real(c_float),      intent(in)     :: xEdges(*), yEdges(*), zEdges(*), totalExt(*), cumExt(*), ssa(*)
      integer(c_int32_t), intent(in)     :: pfIndex(*)
      integer(c_int)                     :: rc
    end function
    function i3rc_hip_destroy(h) bind(C, name = "i3rc_hip_destroy") result(rc)
      import
      type(c_ptr), value :: h
      integer(c_int)     :: rc
    end function
    function i3rc_hip_last_error(h) bind(C, name = "i3rc_hip_last_error") result(text)
      import
      type(c_ptr), value :: h
      type(c_ptr)        :: text
    end function
    function i3rc_hip_set_inverse_table(h, comp, nSteps, nEntries, t) bind(C, name = "i3rc_hip_set_inverse_table") result(rc)
      import
      type(c_ptr), value         :: h
      integer(c_int), value      :: comp, nSteps, nEntries
      real(c_float), intent(in)  :: t(*)
      integer(c_int)             :: rc
    end function
    function i3rc_hip_set_forward_tables(h, comp, nSteps, nEntries, hybrid, orig) &
             bind(C, name = "i3rc_hip_set_forward_tables") result(rc)
      import
      type(c_ptr), value         :: h
      integer(c_int), value      :: comp, nSteps, nEntries
      real(c_float), intent(in)  :: hybrid(*), orig(*)
      integer(c_int)             :: rc
    end function
    function i3rc_hip_set_params(h, p) bind(C, name = "i3rc_hip_set_params") result(rc)
      import
      type(c_ptr), value            :: h
      type(i3rc_params), intent(in) :: p
      integer(c_int)                :: rc
    end function
    function i3rc_hip_set_surface(h, nxs, nys, xs, ys, brdf) bind(C, name = "i3rc_hip_set_surface") result(rc)
      import
      type(c_ptr), value        :: h
      integer(c_int), value     :: nxs, nys
      real(c_float), intent(in) :: xs(*), ys(*), brdf(*)
      integer(c_int)            :: rc
    end function
    function i3rc_hip_set_directions(h, nDir, dirCos) bind(C, name = "i3rc_hip_set_directions") result(rc)
      import
      type(c_ptr), value        :: h
      integer(c_int), value     :: nDir
      real(c_float), intent(in) :: dirCos(*)
      integer(c_int)            :: rc
    end function
    function i3rc_hip_get_tally_layout(h, layout) bind(C, name = "i3rc_hip_get_tally_layout") result(rc)
      import
      type(c_ptr), value                   :: h
      type(i3rc_tally_layout), intent(out) :: layout
      integer(c_int)                       :: rc
    end function
    function i3rc_hip_zero_tallies(h) bind(C, name = "i3rc_hip_zero_tallies") result(rc)
      import
      type(c_ptr), value :: h
      integer(c_int)     :: rc
    end function
    function i3rc_hip_launch_batch(h, seed0, seed1, firstPhoton, nPhotons, src) &
             bind(C, name = "i3rc_hip_launch_batch") result(rc)
      import
      type(c_ptr), value            :: h
      integer(c_int32_t), value     :: seed0, seed1          ! same 32 bits as the C side's uint32_t
      integer(c_int64_t), value     :: firstPhoton, nPhotons
      type(i3rc_source), intent(in) :: src
      integer(c_int)                :: rc
    end function
    function i3rc_hip_compute_batch(h, seed0, seed1, nPhotons, src, lookAhead, hostTallies) &
             bind(C, name = "i3rc_hip_compute_batch") result(rc)
      import
      type(c_ptr), value            :: h
      integer(c_int32_t), value     :: seed0, seed1
      integer(c_int64_t), value     :: nPhotons
      type(i3rc_source), intent(in) :: src
      integer(c_int), value         :: lookAhead            ! batches (seed1 + 1, ...) launched behind this one once a loop shows
      real(c_double), intent(out)   :: hostTallies(*)       ! layout%total
      integer(c_int)                :: rc
    end function
    function i3rc_hip_run_batches(h, seed0, seed1, nBatches, nPhotons, src, inFlight, hostTallies) &
             bind(C, name = "i3rc_hip_run_batches") result(rc)
      import
      type(c_ptr), value            :: h
      integer(c_int32_t), value     :: seed0, seed1          ! batch k is traced with the key (seed0, seed1 + k)
      integer(c_int), value         :: nBatches, inFlight
      integer(c_int64_t), value     :: nPhotons
      type(i3rc_source), intent(in) :: src
      real(c_double), intent(out)   :: hostTallies(*)        ! nBatches * layout%total
      integer(c_int)                :: rc
    end function
    function i3rc_hip_get_moments_layout(h, layout) bind(C, name = "i3rc_hip_get_moments_layout") result(rc)
      import
      type(c_ptr), value                     :: h
      type(i3rc_moments_layout), intent(out) :: layout
      integer(c_int)                         :: rc
    end function
    function i3rc_hip_run_batches_moments(h, seed0, seed1, nBatches, nPhotons, src, total, totalSquares, counters) &
             bind(C, name = "i3rc_hip_run_batches_moments") result(rc)
      import
      type(c_ptr), value            :: h
      integer(c_int32_t), value     :: seed0, seed1          ! batch k is traced with the key (seed0, seed1 + k)
      integer(c_int), value         :: nBatches
      integer(c_int64_t), value     :: nPhotons
      type(i3rc_source), intent(in) :: src
      real(c_double), intent(out)   :: total(*), totalSquares(*)   ! i3rc_moments_layout%total each
      real(c_double), intent(out)   :: counters(*)                 ! I3RC_NUM_COUNTERS
      integer(c_int)                :: rc
    end function
    function i3rc_hip_expect_batches(h, seed0, seed1, nBatches, nPhotons, src, accepted) &
             bind(C, name = "i3rc_hip_expect_batches") result(rc)
      import
      type(c_ptr), value            :: h
      integer(c_int32_t), value     :: seed0, seed1          ! the caller will ask for the batches (seed0, seed1) ... (seed0, seed1 + nBatches - 1)
      integer(c_int), value         :: nBatches
      integer(c_int64_t), value     :: nPhotons
      type(i3rc_source), intent(in) :: src
      integer(c_int), intent(out)   :: accepted              ! 1: traced ahead in fused groups; 0: nothing done
      integer(c_int)                :: rc
    end function
    function i3rc_hip_fetch_tallies(h, host) bind(C, name = "i3rc_hip_fetch_tallies") result(rc)
      import
      type(c_ptr), value          :: h
      real(c_double), intent(out) :: host(*)
      integer(c_int)              :: rc
    end function
    function i3rc_hip_normalise(h, host, fluxUp, fluxDown, fluxAbsorbed, volumeAbsorption, intensity, intensityByComponent) &
             bind(C, name = "i3rc_hip_normalise") result(rc)
      import
      type(c_ptr), value         :: h
      real(c_double), intent(in) :: host(*)
      type(c_ptr), value         :: fluxUp, fluxDown, fluxAbsorbed, volumeAbsorption, intensity, intensityByComponent
      integer(c_int)             :: rc
    end function
    function i3rc_hip_device_count() bind(C, name = "i3rc_hip_device_count") result(n)
      import
      integer(c_int) :: n
    end function
    function c_strlen(s) bind(C, name = "strlen") result(n)
      import
      type(c_ptr), value :: s
      integer(c_size_t)  :: n
    end function
  end interface
contains
  ! text of the last error of handle h (or of the last failed create when h is null)
  function lastErrorText(h) result(text)
    type(c_ptr), intent(in) :: h
    character(len = 256)    :: text
    type(c_ptr) :: p
    character(kind = c_char), pointer :: chars(:)
    integer :: n, i
    text = ""
    p = i3rc_hip_last_error(h)
    if(.not. c_associated(p)) return
    n = int(c_strlen(p))
    call c_f_pointer(p, chars, (/ n /))
    do i = 1, min(n, len(text))
      text(i:i) = chars(i)
    end do
  end function lastErrorText
end module i3rcHipInterface
