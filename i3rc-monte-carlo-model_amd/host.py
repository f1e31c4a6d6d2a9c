"""Python mirror of the reference's Fortran module API for the photon-tracing path, over the C ABI
(include/i3rc_hip.h).  Same names, argument meaning and error behaviour as

  Code/opticalProperties.f95        domain, new_Domain, addOpticalComponent, getInfo_Domain,
                                    getOpticalPropertiesByComponent
  Code/surfaceProperties.f95        surfaceDescription, new_SurfaceDescription
  Code/monteCarloIllumination.f95   photonStream, new_PhotonStream, morePhotonsExist
  Code/RandomNumbersForMC.f95       randomNumberSequence, new_RandomNumberSequence(seed=(/i, j/))
  Integrators/monteCarloRadiativeTransfer.f95
                                    integrator, new_Integrator, specifyParameters, computeRadiativeTransfer,
                                    reportResults, isReady_Integrator, finalize_Integrator

so tests read like the reference's drivers.  Errors that the reference reports through type(ErrorMessage)
are raised as I3RCError with the reference's message text.  The production host is the Fortran shell
(fortran/); this mirror exists because the test and bench harness is Python.
"""
import ctypes as C

import numpy as np

from . import binding as B
from .binding import I3RCError, f32, pf
from .phasefunctions import PI_MCRT, PhaseFunction, PhaseFunctionTable, hybrid_phase_functions, libm_f32, spacing

r32 = np.float32  # scalar float32 (f32() from the binding makes contiguous ARRAYS)

DEFAULT_MIN_TABLE_SIZE = 9001  # monteCarloRadiativeTransfer.f95:36-37


# ---------------------------------------------------------------------------------------------------------
class RandomNumberSequence:
    """new_RandomNumberSequence(seed): on the GPU the seed pair keys the per-photon Philox streams."""

    def __init__(self, seed):
        if np.isscalar(seed):
            seed = (int(seed), 0)
        seed = tuple(int(s) & 0xFFFFFFFF for s in seed)
        if len(seed) != 2:
            raise I3RCError("new_RandomNumberSequence: the GPU integrator takes seed=(/i, j/)")
        self.seed = seed
        self.photonsDrawn = 0   # photons whose Philox streams this sequence has handed out (see Integrator.launch)


class PhotonStream:
    """new_PhotonStream: Directional streams (Code/monteCarloIllumination.f95:62-104) are represented lazily
    and generated on the device; any other illumination can be passed as explicit arrays."""

    def __init__(self, solarMu=None, solarAzimuth=None, numberOfPhotons=0, randomNumbers=None, arrays=None):
        if numberOfPhotons <= 0 and arrays is None:
            raise I3RCError("setIllumination: must ask for non-negative number of photons.")
        if arrays is not None:
            self.kind = 1
            self.arrays = [f32(a) for a in arrays]
            self.n = len(self.arrays[0])
            if any(len(a) != self.n for a in self.arrays) or len(self.arrays) != 5:
                raise I3RCError("setIllumination: need x, y, z, mu, phi arrays of equal length")
        else:
            if solarAzimuth < 0.0 or solarAzimuth > 360.0:
                raise I3RCError("setIllumination: solarAzimuth out of bounds")
            if abs(solarMu) > 1.0 or abs(solarMu) <= np.finfo(np.float32).tiny:
                raise I3RCError("setIllumination: solarMu out of bounds")
            self.kind = 0
            self.solarMu, self.solarAzimuth = float(solarMu), float(solarAzimuth)
            self.n = int(numberOfPhotons)
        self.currentPhoton = 1

    def morePhotonsExist(self):
        return 0 < self.currentPhoton <= self.n


# ---------------------------------------------------------------------------------------------------------
class SurfaceDescription:
    """new_SurfaceDescription(surfaceParameters[, xPosition, yPosition]) (Code/surfaceProperties.f95:60-117)."""

    def __init__(self, surfaceParameters, xPosition=None, yPosition=None):
        p = np.asarray(surfaceParameters, np.float32)
        if xPosition is None:  # newSurfaceUniform :98-117
            if p.size != 1:
                raise I3RCError("new_SurfaceDescription: Wrong number of parameters supplied for surface BRDF.")
            huge = np.finfo(np.float32).max
            self.x = np.array([0.0, huge], np.float32)
            self.y = np.array([0.0, huge], np.float32)
            self.albedo = p.reshape(1, 1)
        else:
            self.x, self.y = f32(xPosition), f32(yPosition)
            if p.ndim == 3:  # (1, nx, ny) Fortran order -> [ny, nx]
                if p.shape[0] != 1:
                    raise I3RCError("new_SurfaceDescription: Wrong number of parameters supplied for surface BRDF.")
                p = p[0].T
            self.albedo = np.ascontiguousarray(p, np.float32)
            if self.albedo.shape != (len(self.y) - 1, len(self.x) - 1):
                raise I3RCError("new_SurfaceDescription: position vector(s) are incorrect length.")
            if np.any(np.diff(self.x) <= 0) or np.any(np.diff(self.y) <= 0):
                raise I3RCError("new_SurfaceDescription: positions must be unique, increasing.")
        if np.any(self.albedo < 0) or np.any(self.albedo > 1):
            raise I3RCError("new_SurfaceDescription: surface reflectance must be between 0 and 1")


# ---------------------------------------------------------------------------------------------------------
class Domain:
    """type(domain) (Code/opticalProperties.f95:54-65).  Arrays are numpy [z, y, x] (x fastest)."""

    def __init__(self, xPosition, yPosition, zPosition):
        self.x, self.y, self.z = f32(xPosition), f32(yPosition), f32(zPosition)
        if np.any(np.diff(self.x) <= 0) or np.any(np.diff(self.y) <= 0) or np.any(np.diff(self.z) <= 0):
            raise I3RCError("new_Domain: Positions must be increasing, unique.")
        self.components = []

    @property
    def shape(self):
        return len(self.z) - 1, len(self.y) - 1, len(self.x) - 1

    def addOpticalComponent(self, componentName, extinction, singleScatteringAlbedo, phaseFunctionIndex,
                            phaseFunctions, zLevelBase=1):
        """addOpticalComponent3D / 1D (:133-230) with validateOpticalComponent's checks (:929-987)."""
        nz, ny, nx = self.shape
        ext = f32(extinction)
        ssa = f32(singleScatteringAlbedo)
        pfi = np.ascontiguousarray(phaseFunctionIndex, np.int32)
        uniform = ext.ndim == 1
        if uniform:
            ext, ssa, pfi = ext[:, None, None], ssa[:, None, None], pfi[:, None, None]
        if ext.shape != ssa.shape or ext.shape != pfi.shape:
            raise I3RCError("validateOpticalComponent: optical property grids must be the same size.")
        if not uniform and ext.shape[1:] != (ny, nx):
            raise I3RCError("validateOpticalComponent: optical property grids don't match the domain in x, y.")
        if zLevelBase < 1 or zLevelBase + ext.shape[0] - 1 > nz:
            raise I3RCError("validateOpticalComponent: optical property grids don't fit the domain in z.")
        if np.any(ext < 0):
            raise I3RCError("validateOpticalComponent: extinction must be >= 0.")
        if np.any(ssa < 0) or np.any(ssa > 1):
            raise I3RCError("validateOpticalComponent: singleScatteringAlbedo must be between 0 and 1")
        if np.any(pfi < 0) or np.any(pfi > phaseFunctions.n_entries):
            raise I3RCError("validateOpticalComponent: phase function index is out of bounds")
        if np.any((pfi == 0) & (ext > 0)):
            raise I3RCError("validateOpticalComponent: phase function index is 0 where extinction is non-zero")
        self.components.append(dict(name=componentName, ext=ext, ssa=ssa, pfi=pfi, table=phaseFunctions,
                                    zbase=int(zLevelBase), uniform=uniform))

    def getInfo_Domain(self):
        nz, ny, nx = self.shape
        return dict(numX=nx, numY=ny, numZ=nz, xPosition=self.x.copy(), yPosition=self.y.copy(),
                    zPosition=self.z.copy(), numberOfComponents=len(self.components),
                    componentNames=[c["name"] for c in self.components])

    def getOpticalPropertiesByComponent(self):
        """:429-539.  Returns totalExt [z,y,x], cumulativeExt / ssa / phaseFunctionIndex [comp,z,y,x], tables."""
        if not self.components:
            raise I3RCError("getOpticalPropertiesByComponent: domain contains no optical components.")
        nz, ny, nx = self.shape
        nc = len(self.components)
        cum = np.zeros((nc, nz, ny, nx), np.float32)
        ssa = np.zeros((nc, nz, ny, nx), np.float32)
        pfi = np.zeros((nc, nz, ny, nx), np.int32)
        for i, c in enumerate(self.components):
            z0, z1 = c["zbase"] - 1, c["zbase"] - 1 + c["ext"].shape[0]
            cum[i, z0:z1] = np.broadcast_to(c["ext"], (z1 - z0, ny, nx))
            ssa[i, z0:z1] = np.broadcast_to(c["ssa"], (z1 - z0, ny, nx))
            pfi[i, z0:z1] = np.broadcast_to(c["pfi"], (z1 - z0, ny, nx))
        for i in range(1, nc):
            cum[i] = cum[i] + cum[i - 1]
        total = cum[-1].copy()
        mask = total > np.finfo(np.float32).tiny
        with np.errstate(all="ignore"):
            cum = np.where(mask[None], cum / np.where(mask, total, r32(1.0))[None], cum).astype(np.float32)
        return total, cum, ssa, pfi, [c["table"] for c in self.components]


# ---------------------------------------------------------------------------------------------------------
class Integrator:
    """type(integrator): owns a device-resident copy of the problem (new_Integrator :162-254)."""

    _PARAM_KEYS = ("surfaceAlbedo", "surfaceBDRF", "minForwardTableSize", "minInverseTableSize", "intensityMus",
                   "intensityPhis", "computeIntensity", "useRayTracing", "useRussianRoulette",
                   "useRussianRouletteForIntensity", "zetaMin", "useHybridPhaseFunsForIntenCalcs",
                   "hybridPhaseFunWidth", "numOrdersOrigPhaseFunIntenCalcs", "limitIntensityContributions",
                   "maxIntensityContribution")

    def __init__(self, atmosphere, device=0):
        self._h = C.c_void_p()
        self._lib = B.load()
        total, cum, ssa, pfi, tables = atmosphere.getOpticalPropertiesByComponent()
        # :233-234: nudge the last cumulative slice so that r == 1 still selects the last component
        one = r32(1.0)
        last = cum[-1]
        last[np.abs(last - one) <= np.spacing(one)] = one + np.spacing(one)
        self.nz, self.ny, self.nx = total.shape
        self.ncomp = cum.shape[0]
        self.x, self.y, self.z = atmosphere.x.copy(), atmosphere.y.copy(), atmosphere.z.copy()
        self.forwardTables = tables
        total, cum, ssa = f32(total), f32(cum), f32(ssa)
        pfi = np.ascontiguousarray(pfi, np.int32)
        rc = self._lib.i3rc_hip_create(C.byref(self._h), int(device), self.nx, self.ny, self.nz, self.ncomp,
                                       pf(self.x), pf(self.y), pf(self.z), pf(total), pf(cum), pf(ssa),
                                       pfi.ctypes.data_as(B.ip))
        if rc != 0:
            msg = self._lib.i3rc_hip_last_error(None).decode()
            self._h = C.c_void_p()
            raise I3RCError("new_Integrator: " + msg)
        self.params = B.Params(0.0, 0, 1, 1, 0, 0, 0, 0.3, 0, float(np.finfo(np.float32).max))
        self.minForwardTableSize = DEFAULT_MIN_TABLE_SIZE
        self.minInverseTableSize = DEFAULT_MIN_TABLE_SIZE
        self.hybridPhaseFunWidth = 7.0
        self.intensityDirections = np.zeros((0, 3), np.float32)
        self.computeIntensity = False
        self._inv_size = [0] * self.ncomp
        self._fwd_size = [0] * self.ncomp
        self._fwd_stale = True
        self._results = None
        self.readyToCompute = True

    # -- helpers
    def _check(self, rc, where):
        if rc != 0:
            raise I3RCError(f"{where}: " + self._lib.i3rc_hip_last_error(self._h).decode())

    def isReady_Integrator(self):
        return bool(self.readyToCompute and self._h)

    def finalize_Integrator(self):
        if self._h:
            self._lib.i3rc_hip_destroy(self._h)
            self._h = C.c_void_p()
        self.readyToCompute = False

    def __del__(self):
        try:
            self.finalize_Integrator()
        except Exception:
            pass

    # -- specifyParameters :830-1069
    def specifyParameters(self, **kw):
        for k in kw:
            if k not in self._PARAM_KEYS:
                raise TypeError(f"specifyParameters: unknown keyword {k}")
        if "surfaceBDRF" in kw and "surfaceAlbedo" in kw:
            raise I3RCError("specifyParameters: only one surface specification can be provided")
        if "surfaceAlbedo" in kw and not (0.0 <= kw["surfaceAlbedo"] <= 1.0):
            raise I3RCError("specifyParameters: surface albedo out of range.")
        if ("intensityMus" in kw) != ("intensityPhis" in kw):
            raise I3RCError("specifyParameters: Both or neither of intensityMus and intensityPhis must be supplied")
        if "intensityMus" in kw:
            mus, phis = f32(kw["intensityMus"]), f32(kw["intensityPhis"])
            if mus.shape != phis.shape:
                raise I3RCError("specifyParameters: intensityMus, intensityPhis must be the same length.")
            if np.any(mus < -1) or np.any(mus > 1):
                raise I3RCError("specifyParameters: intensityMus must be between -1 and 1")
            if np.any(np.abs(mus) < np.finfo(np.float32).tiny):
                raise I3RCError("specifyParameters: intensityMus can't be 0 (directly sideways)")
            if np.any(phis < 0) or np.any(phis > 360):
                raise I3RCError("specifyParameters: intensityPhis must be between 0 and 360")
        if kw.get("computeIntensity") and "intensityMus" not in kw and len(self.intensityDirections) == 0:
            raise I3RCError("specifyParameters: Can't compute intensity without specifying directions.")

        p = self.params
        if "surfaceAlbedo" in kw:
            p.surfaceAlbedo, p.useSurfaceBDRF = float(kw["surfaceAlbedo"]), 0
        elif "surfaceBDRF" in kw:
            s = kw["surfaceBDRF"]
            self._check(self._lib.i3rc_hip_set_surface(self._h, len(s.x) - 1, len(s.y) - 1, pf(s.x), pf(s.y), pf(s.albedo)),
                        "specifyParameters")
            p.useSurfaceBDRF = 1
        if "useRayTracing" in kw:
            p.useRayTracing = int(bool(kw["useRayTracing"]))
        if "minForwardTableSize" in kw:
            self.minForwardTableSize = max(int(kw["minForwardTableSize"]), DEFAULT_MIN_TABLE_SIZE)
        if "minInverseTableSize" in kw:
            self.minInverseTableSize = max(int(kw["minInverseTableSize"]), DEFAULT_MIN_TABLE_SIZE)
        if "useRussianRoulette" in kw:
            p.useRussianRoulette = int(bool(kw["useRussianRoulette"]))
        if "useRussianRouletteForIntensity" in kw:
            p.useRussianRouletteForIntensity = int(bool(kw["useRussianRouletteForIntensity"]))
        if "zetaMin" in kw and kw["zetaMin"] >= 0:
            p.zetaMin = float(kw["zetaMin"])
        if "useHybridPhaseFunsForIntenCalcs" in kw:
            p.useHybridPhaseFunsForIntenCalcs = int(bool(kw["useHybridPhaseFunsForIntenCalcs"]))
            self._fwd_stale = True
        if "hybridPhaseFunWidth" in kw:
            wd = float(kw["hybridPhaseFunWidth"])
            self.hybridPhaseFunWidth = wd if 0 < wd < 30.0 else 7.0
            self._fwd_stale = True
        if "numOrdersOrigPhaseFunIntenCalcs" in kw:
            n = int(kw["numOrdersOrigPhaseFunIntenCalcs"])
            p.numOrdersOrigPhaseFunIntenCalcs = n if n >= 0 else 0
        if "limitIntensityContributions" in kw:
            p.limitIntensityContributions = int(bool(kw["limitIntensityContributions"]))
        if kw.get("maxIntensityContribution", 0) > 0:
            p.maxIntensityContribution = float(kw["maxIntensityContribution"])
        if "intensityMus" in kw:
            d = []
            for m, ph in zip(f32(kw["intensityMus"]), f32(kw["intensityPhis"])):
                phr = r32(r32(ph * PI_MCRT) / r32(180.0))
                st = np.sqrt(r32(1.0) - m * m, dtype=np.float32)
                d.append([st * libm_f32("cosf", phr), st * libm_f32("sinf", phr), m])  # :2041-2059 (libm: what the reference's cos / sin are)
            dirs = np.array(d, np.float32).reshape(-1, 3)
            self._check(self._lib.i3rc_hip_set_directions(self._h, len(d), pf(dirs)), "specifyParameters")
            self.intensityDirections = dirs      # (only once the device has taken them: a refusal leaves the object as it was)
            self.computeIntensity = True
        if "computeIntensity" in kw and not kw["computeIntensity"] and "intensityMus" not in kw:
            self._check(self._lib.i3rc_hip_set_directions(self._h, 0, None), "specifyParameters")
            self.intensityDirections = np.zeros((0, 3), np.float32)
            self.computeIntensity = False
        self._check(self._lib.i3rc_hip_set_params(self._h, C.byref(p)), "specifyParameters")

    # -- tabulateInversePhaseFunctions :1809-1861, tabulateForwardPhaseFunctions :1863-1923
    def _ensure_tables(self):
        for c in range(self.ncomp):
            if self._inv_size[c] < self.minInverseTableSize:
                t = f32(self.forwardTables[c].inverse_table(self.minInverseTableSize))
                self._check(self._lib.i3rc_hip_set_inverse_table(self._h, c + 1, t.shape[1], t.shape[0], pf(t)),
                            "tabulateInversePhaseFunctions")
                self._inv_size[c] = t.shape[1]
            if self.computeIntensity and (self._fwd_size[c] < self.minForwardTableSize or self._fwd_stale):
                orig = f32(self.forwardTables[c].forward_table(self.minForwardTableSize))
                hyb = orig
                if self.params.useHybridPhaseFunsForIntenCalcs and self.hybridPhaseFunWidth > 0:
                    hyb = f32(hybrid_phase_functions(orig, self.hybridPhaseFunWidth))
                self._check(self._lib.i3rc_hip_set_forward_tables(self._h, c + 1, orig.shape[1], orig.shape[0], pf(hyb), pf(orig)),
                            "tabulateForwardPhaseFunctions")
                self._fwd_size[c] = orig.shape[1]
        self._fwd_stale = False

    def set_tables(self, component, inverse=None, forward=None, forward_orig=None):
        """Hand tables in directly (tests: golden tables from the oracle)."""
        if inverse is not None:
            t = f32(np.atleast_2d(inverse))
            self._check(self._lib.i3rc_hip_set_inverse_table(self._h, component, t.shape[1], t.shape[0], pf(t)), "set_tables")
            self._inv_size[component - 1] = max(t.shape[1], 10 ** 9)
        if forward is not None:
            t = f32(np.atleast_2d(forward))
            o = t if forward_orig is None else f32(np.atleast_2d(forward_orig))
            self._check(self._lib.i3rc_hip_set_forward_tables(self._h, component, t.shape[1], t.shape[0], pf(t), pf(o)), "set_tables")
            self._fwd_size[component - 1] = max(t.shape[1], 10 ** 9)
            self._fwd_stale = False          # (tables handed over ready-made are the ones to use)

    # -- computeRadiativeTransfer :262-398
    def launch(self, randomNumbers, incomingPhotons, firstPhoton=None, zero=True):
        """Asynchronous part: zero the tallies (as the reference does per call) and launch the batch.
        Photon i of a sequence has the Philox stream (key = seed, counter = i).  A sequence that is used again without
        being re-seeded goes on with the next photon numbers -- the reference's Mersenne Twister simply goes on too --
        unless the caller names the first photon itself (sharding a batch over ranks)."""
        if not self.isReady_Integrator():
            raise I3RCError("computeRadiativeTransfer: problem not completely specified.")
        if not incomingPhotons.morePhotonsExist():
            raise I3RCError("computeRadiativeTransfer: Didn't process any photons.")
        self._ensure_tables()
        if zero:
            self._check(self._lib.i3rc_hip_zero_tallies(self._h), "computeRadiativeTransfer")
        s = B.Source()
        s.kind = incomingPhotons.kind
        if s.kind == 0:
            s.solarMu, s.solarAzimuth = incomingPhotons.solarMu, incomingPhotons.solarAzimuth
        else:
            s.x, s.y, s.z, s.mu, s.phi = [pf(a) for a in incomingPhotons.arrays]
        if firstPhoton is None:
            firstPhoton = randomNumbers.photonsDrawn
            randomNumbers.photonsDrawn += incomingPhotons.n
        self._check(self._lib.i3rc_hip_launch_batch(self._h, randomNumbers.seed[0], randomNumbers.seed[1], int(firstPhoton),
                                                    incomingPhotons.n, C.byref(s)), "computeRadiativeTransfer")
        incomingPhotons.currentPhoton = incomingPhotons.n + 1  # the stream is consumed

    def layout(self):
        lay = B.TallyLayout()
        self._check(self._lib.i3rc_hip_get_tally_layout(self._h, C.byref(lay)), "layout")
        return lay

    def fetch(self):
        lay = self.layout()
        raw = np.zeros(lay.total, np.float64)
        self._check(self._lib.i3rc_hip_fetch_tallies(self._h, raw.ctypes.data_as(B.dp)), "computeRadiativeTransfer")
        return raw

    def finish(self, raw=None):
        """Synchronise, fetch raw tallies, normalise (:353-395) and cache the results for reportResults."""
        raw = self.fetch() if raw is None else raw
        lay = self.layout()
        nd = len(self.intensityDirections)
        res = dict(fluxUp=np.zeros((self.ny, self.nx), np.float32), fluxDown=np.zeros((self.ny, self.nx), np.float32),
                   fluxAbsorbed=np.zeros((self.ny, self.nx), np.float32),
                   volumeAbsorption=np.zeros((self.nz, self.ny, self.nx), np.float32))
        inten = np.zeros((max(nd, 1), self.ny, self.nx), np.float32)
        byc = np.zeros((self.ncomp + 1, max(nd, 1), self.ny, self.nx), np.float32)
        self._check(self._lib.i3rc_hip_normalise(self._h, raw.ctypes.data_as(B.dp), pf(res["fluxUp"]), pf(res["fluxDown"]),
                                                 pf(res["fluxAbsorbed"]), pf(res["volumeAbsorption"]), pf(inten), pf(byc)),
                    "computeRadiativeTransfer")
        if nd:
            res["intensity"], res["intensityByComponent"] = inten, byc
        cnt = raw[lay.counters:lay.counters + B.NUM_COUNTERS]
        res["counters"] = {k: float(cnt[i]) for i, k in enumerate(B.COUNTER_NAMES)}
        res["raw"] = raw
        if res["counters"]["photons"] <= 0:
            raise I3RCError("computeRadiativeTransfer: Didn't process any photons.")
        self._results = res
        return res

    def computeRadiativeTransfer(self, randomNumbers, incomingPhotons):
        self.launch(randomNumbers, incomingPhotons)
        return self.finish()

    def computeRadiativeTransferLookingAhead(self, randomNumbers, incomingPhotons, lookAhead=3):
        """computeRadiativeTransfer through i3rc_hip_compute_batch: one batch of a driver's loop, with the following batches
        launched ahead once the loop shows (same batch, next seed word).  Directional streams of a fresh sequence only."""
        if not self.isReady_Integrator():
            raise I3RCError("computeRadiativeTransfer: problem not completely specified.")
        if incomingPhotons.kind != 0 or randomNumbers.photonsDrawn != 0:
            return self.computeRadiativeTransfer(randomNumbers, incomingPhotons)
        if not incomingPhotons.morePhotonsExist():
            raise I3RCError("computeRadiativeTransfer: Didn't process any photons.")
        self._ensure_tables()
        raw = np.zeros(self.layout().total, np.float64)
        s = B.Source()
        s.kind, s.solarMu, s.solarAzimuth = 0, incomingPhotons.solarMu, incomingPhotons.solarAzimuth
        self._check(self._lib.i3rc_hip_compute_batch(self._h, randomNumbers.seed[0], randomNumbers.seed[1], incomingPhotons.n,
                                                     C.byref(s), int(lookAhead), raw.ctypes.data_as(B.dp)), "computeRadiativeTransfer")
        randomNumbers.photonsDrawn += incomingPhotons.n
        incomingPhotons.currentPhoton = incomingPhotons.n + 1
        return self.finish(raw)

    def computeRadiativeTransferBatches(self, seed, numBatches, solarMu, solarAzimuth, numberOfPhotons, inFlight=0):
        """A driver's batch loop (monteCarloDriver.f95:283-326) as one call of the C ABI (i3rc_hip_run_batches): batch k
        is what computeRadiativeTransfer(new_RandomNumberSequence((seed[0], seed[1] + k)), new_PhotonStream(solarMu,
        solarAzimuth, numberOfPhotons)) gives, but up to inFlight batches share the device, so that one batch's long
        tail is covered by the next.  Returns the list of per-batch results (as finish() makes them)."""
        if not self.isReady_Integrator():
            raise I3RCError("computeRadiativeTransfer: problem not completely specified.")
        self._ensure_tables()
        lay = self.layout()
        raw = np.zeros((int(numBatches), lay.total), np.float64)
        s = B.Source()
        s.kind, s.solarMu, s.solarAzimuth = 0, solarMu, solarAzimuth
        self._check(self._lib.i3rc_hip_run_batches(self._h, int(seed[0]), int(seed[1]), int(numBatches), int(numberOfPhotons),
                                                   C.byref(s), int(inFlight), raw.ctypes.data_as(B.dp)), "computeRadiativeTransfer")
        return [self.finish(raw[k]) for k in range(int(numBatches))]

    def computeRadiativeTransferBatchMoments(self, seed, numBatches, solarMu, solarAzimuth, numberOfPhotons):
        """A driver's batch loop with its statistics gathered on the device (i3rc_hip_run_batches_moments): the sums and sums of
        squares over the batches of everything reportResults hands out (monteCarloDriver.f95:300-321), as two dicts of arrays in
        reportResults' shapes, plus the work counters summed over the batches.  No per-batch block comes back to the host."""
        if not self.isReady_Integrator():
            raise I3RCError("computeRadiativeTransfer: problem not completely specified.")
        self._ensure_tables()
        lay = B.MomentsLayout()
        self._check(self._lib.i3rc_hip_get_moments_layout(self._h, C.byref(lay)), "computeRadiativeTransfer")
        s1, s2, cnt = np.zeros(lay.total, np.float64), np.zeros(lay.total, np.float64), np.zeros(B.NUM_COUNTERS, np.float64)
        s = B.Source()
        s.kind, s.solarMu, s.solarAzimuth = 0, solarMu, solarAzimuth
        self._check(self._lib.i3rc_hip_run_batches_moments(self._h, int(seed[0]), int(seed[1]), int(numBatches), int(numberOfPhotons), C.byref(s),
                                                           s1.ctypes.data_as(B.dp), s2.ctypes.data_as(B.dp), cnt.ctypes.data_as(B.dp)), "computeRadiativeTransfer")
        nd, ncol = len(self.intensityDirections), self.nx * self.ny

        def unpack(m):
            out = dict(fluxUp=m[lay.fluxUp:lay.fluxUp + ncol].reshape(self.ny, self.nx), fluxDown=m[lay.fluxDown:lay.fluxDown + ncol].reshape(self.ny, self.nx),
                       fluxAbsorbed=m[lay.fluxAbsorbed:lay.fluxAbsorbed + ncol].reshape(self.ny, self.nx),
                       volumeAbsorption=m[lay.volumeAbsorption:lay.volumeAbsorption + ncol * self.nz].reshape(self.nz, self.ny, self.nx),
                       absorbedProfile=m[lay.absorbedProfile:lay.absorbedProfile + self.nz], meanFluxUp=m[lay.meanFluxUp],
                       meanFluxDown=m[lay.meanFluxDown], meanFluxAbsorbed=m[lay.meanFluxAbsorbed])
            if nd:
                out["intensity"] = m[lay.intensity:lay.intensity + nd * ncol].reshape(nd, self.ny, self.nx)
                out["meanIntensity"] = m[lay.meanIntensity:lay.meanIntensity + nd]
            return out
        return unpack(s1), unpack(s2), {k: float(cnt[i]) for i, k in enumerate(B.COUNTER_NAMES)}

    def set_batch_fusion(self, mode):
        """-1: automatic (default), 0: every batch a launch of its own, 1: fuse a loop's batches into one grid whenever the
        problem allows (i3rc_hip_set_batch_fusion)."""
        self._check(self._lib.i3rc_hip_set_batch_fusion(self._h, int(mode)), "set_batch_fusion")

    def set_lds_tallies(self, on):
        """False: a plain launch adds every tally straight to the float64 buffer in global memory (i3rc_hip_set_lds_tallies)."""
        self._check(self._lib.i3rc_hip_set_lds_tallies(self._h, 1 if on else 0), "set_lds_tallies")

    def kernel_ms(self):
        ms = C.c_float(0)
        self._check(self._lib.i3rc_hip_last_kernel_ms(self._h, C.byref(ms)), "kernel_ms")
        return float(ms.value)

    def timed_launches(self):
        return int(self._lib.i3rc_hip_timed_launch_count(self._h))

    def kernel_name(self):
        if not hasattr(self._lib, "i3rc_hip_last_kernel_name"):
            return "?"
        return self._lib.i3rc_hip_last_kernel_name(self._h).decode()

    def kernel_ms_history(self, n):
        ms = np.zeros(n, np.float32)
        self._check(self._lib.i3rc_hip_kernel_ms_history(self._h, int(n), pf(ms)), "kernel_ms_history")
        return ms

    # -- reportResults :711-826
    def reportResults(self):
        r = self._results
        if r is None:
            raise I3RCError("reportResults: no results available")
        ncol = r32(self.nx * self.ny)
        out = dict(meanFluxUp=r["fluxUp"].sum(dtype=np.float32) / ncol, meanFluxDown=r["fluxDown"].sum(dtype=np.float32) / ncol,
                   meanFluxAbsorbed=r["fluxAbsorbed"].sum(dtype=np.float32) / ncol,
                   fluxUp=r["fluxUp"], fluxDown=r["fluxDown"], fluxAbsorbed=r["fluxAbsorbed"],
                   absorbedProfile=r["volumeAbsorption"].sum(axis=(1, 2), dtype=np.float32) / ncol,
                   volumeAbsorption=r["volumeAbsorption"])
        if "intensity" in r:
            out["intensity"] = r["intensity"]
            out["meanIntensity"] = r["intensity"].sum(axis=(1, 2), dtype=np.float32) / ncol
        return out

    # -- test hooks
    KERNELS = {"auto": 0, "general": 1, "lane": 2, "ring": 3}

    def set_tuning(self, evThreshold=0, blocksPerCU=0, forceGeneral=None, kernel=None, lightThreshold=None):
        """Experiment knobs of the C ABI (i3rc_hip_set_tuning / i3rc_hip_select_kernel); kernel is one of KERNELS."""
        self._check(self._lib.i3rc_hip_set_tuning(self._h, int(evThreshold), int(blocksPerCU)), "set_tuning")
        if forceGeneral is not None:
            self._check(self._lib.i3rc_hip_force_general_kernel(self._h, int(bool(forceGeneral))), "set_tuning")
        if kernel is not None:
            self._check(self._lib.i3rc_hip_select_kernel(self._h, self.KERNELS[kernel]), "set_tuning")
        if lightThreshold is not None:
            self._check(self._lib.i3rc_hip_set_light_threshold(self._h, int(lightThreshold)), "set_tuning")

    GRID_PLACES = {"auto": 0, "linear": 1, "bricks": 2, "columns": 3}

    def select_grid_place(self, place):
        """Where the kernels read the extinction field from (i3rc_hip_select_grid_place); place is one of GRID_PLACES."""
        self._check(self._lib.i3rc_hip_select_grid_place(self._h, self.GRID_PLACES[place]), "select_grid_place")

    def has_column_records(self):
        return bool(self._lib.i3rc_hip_has_column_records(self._h))

    def trace_rays(self, direction, pos, idx, target=None):
        d, p = f32(direction).reshape(-1, 3).copy(), f32(pos).reshape(-1, 3).copy()
        i = np.ascontiguousarray(idx, np.int32).reshape(-1, 3).copy()
        n = len(d)
        t = np.full(n, -1.0, np.float32) if target is None else f32(target).copy()
        tau, steps = np.zeros(n, np.float32), np.zeros(n, np.int32)
        self._check(self._lib.i3rc_hip_trace_rays(self._h, n, pf(d), pf(p), i.ctypes.data_as(B.ip), pf(t), pf(tau),
                                                  steps.ctypes.data_as(B.ip)), "trace_rays")
        return tau, p, i, steps

    def run_replay(self, incomingPhotons, randoms, drawStart):
        self._ensure_tables()
        self._check(self._lib.i3rc_hip_zero_tallies(self._h), "run_replay")
        n = incomingPhotons.n
        s = B.Source()
        s.kind = 1
        s.x, s.y, s.z, s.mu, s.phi = [pf(a) for a in incomingPhotons.arrays]
        randoms = f32(randoms)
        ds = np.ascontiguousarray(drawStart[:n], np.int64)
        out = dict(fate=np.zeros(n, np.int32), fateColumn=np.zeros(n, np.int32), fateWeight=np.zeros(n, np.float32),
                   fateOrder=np.zeros(n, np.int32), drawsUsed=np.zeros(n, np.int32))
        self._check(self._lib.i3rc_hip_run_replay(self._h, n, C.byref(s), pf(randoms), len(randoms), ds.ctypes.data_as(B.lp),
                                                  out["fate"].ctypes.data_as(B.ip), out["fateColumn"].ctypes.data_as(B.ip),
                                                  pf(out["fateWeight"]), out["fateOrder"].ctypes.data_as(B.ip),
                                                  out["drawsUsed"].ctypes.data_as(B.ip)), "run_replay")
        out.update(self.finish())
        return out

    def arith_check(self, num, den):
        num, den = f32(num), f32(den)
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.i3rc_hip_arith_check(self._h, len(num), pf(num), pf(den), C.byref(a), C.byref(b)), "arith_check")
        return a.value, b.value

    def find_index(self, values, table, firstGuess=None):
        """Test hook: findIndex on the device, as the kernels evaluate it (i3rc_hip_find_index)."""
        values, table = f32(values), f32(table)
        g = None if firstGuess is None else np.ascontiguousarray(firstGuess, np.int32)
        out = np.zeros(len(values), np.int32)
        self._check(self._lib.i3rc_hip_find_index(self._h, len(table), pf(table), len(values), pf(values),
                                                  None if g is None else g.ctypes.data_as(B.ip), out.ctypes.data_as(B.ip)), "find_index")
        return out

    def surface_reflectance(self, x, y):
        """Test hook: computeSurfaceReflectance on the device for the surface of specifyParameters(surfaceBDRF=) (i3rc_hip_surface_reflectance)."""
        x, y = f32(x), f32(y)
        out = np.zeros(len(x), np.float32)
        self._check(self._lib.i3rc_hip_surface_reflectance(self._h, len(x), pf(x), pf(y), pf(out)), "surface_reflectance")
        return out

    def philox_blocks(self, seed, firstPhoton, n, blocks):
        out = np.zeros((n, blocks, 4), np.uint32)
        outf = np.zeros((n, blocks, 4), np.float32)
        self._check(self._lib.i3rc_hip_philox_blocks(self._h, seed[0], seed[1], firstPhoton, n, blocks,
                                                     out.ctypes.data_as(B.up), pf(outf)), "philox_blocks")
        return out, outf


# Reference-style free functions ---------------------------------------------------------------------------
def new_Domain(xPosition, yPosition, zPosition):
    return Domain(xPosition, yPosition, zPosition)


def read_Domain(fileName):
    """read_Domain (Code/opticalProperties.f95:708-871) and, per component, read_PhaseFunctionTable
    (Code/scatteringPhaseFunctions.f95:928-1090): a domain file -- netCDF classic, the schema of SURVEY.md 8f row 1 --
    into a Domain.  Arrays in the file are (z, y, x), a horizontally uniform component is (z); the storage type of a
    component's table is "LegendreCoefficients" (start / length vectors into one coefficient vector) or "Angle-Value"."""
    from scipy.io import netcdf_file

    try:
        f = netcdf_file(fileName, "r", mmap=False)
    except (OSError, TypeError, ValueError) as e:
        raise I3RCError(f"read_Domain: Can't open file {fileName}: {e}")
    try:
        var = lambda n: np.array(f.variables[n].data)   # noqa: E731
        text = lambda a: (a.decode() if isinstance(a, bytes) else str(a)).strip()   # noqa: E731
        d = Domain(var("x-Edges"), var("y-Edges"), var("z-Edges"))
        for c in range(1, int(f.numberOfComponents) + 1):
            pre = f"Component{c}_"
            if text(getattr(f, pre + "phaseFunctionStorageType")) == "Angle-Value":
                ang, val = var(pre + "scatteringAngle"), var(pre + "phaseFunctionValues")   # (entry, angle)
                entries = [PhaseFunction(angles=ang, values=v) for v in val.reshape(-1, ang.size)]
            else:
                coef, start, length = var(pre + "legendreCoefficients"), var(pre + "start"), var(pre + "length")
                entries = [PhaseFunction(legendre=coef[s - 1:s - 1 + n]) for s, n in zip(start, length)]
            table = PhaseFunctionTable(entries, key=var(pre + "phaseFunctionKeyT"),
                                       description=text(getattr(f, pre + "description", b"")))
            d.addOpticalComponent(text(getattr(f, pre + "Name")), var(pre + "Extinction"), var(pre + "SingleScatteringAlbedo"),
                                  var(pre + "PhaseFunctionIndex").astype(np.int32), table, zLevelBase=int(getattr(f, pre + "zLevelBase")))
        return d
    except (KeyError, AttributeError) as e:
        raise I3RCError(f"read_Domain: {fileName} doesn't appear to be a domain file: {e}")


def new_Integrator(atmosphere, device=0):
    return Integrator(atmosphere, device)


def new_PhotonStream(solarMu, solarAzimuth, numberOfPhotons, randomNumbers=None):
    return PhotonStream(solarMu, solarAzimuth, numberOfPhotons, randomNumbers)


def new_RandomNumberSequence(seed):
    return RandomNumberSequence(seed)


def new_SurfaceDescription(surfaceParameters, xPosition=None, yPosition=None):
    return SurfaceDescription(surfaceParameters, xPosition, yPosition)


def new_PhaseFunctionTable(phaseFunctions, key=None, tableDescription=""):
    return PhaseFunctionTable(phaseFunctions, key, tableDescription)
