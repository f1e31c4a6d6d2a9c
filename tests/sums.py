"""What "the same photons give the same tallies" means, as a number.

Every tally word is a sum of float32 increments (a photon's weight, a ray's contribution) added up in FLOAT64 all the way --
the partial sums a workgroup gathers in LDS included (csrc/tracer.hpp, tally_t; until round 4 those were float32 and these
comparisons had to allow 1e-5 ... 3e-4, enough to hide a small real difference).  Two runs of the same photons -- another launch
geometry, a fused launch, a look-ahead group, a sharded batch, another place of the extinction field -- therefore differ by the
ORDER of float64 additions only: a sum of N terms moves by at most (N - 1) 2^-53 of the sum of their magnitudes, and no word
can have received more additions than the run had tally events (its own work counters say how many)."""
import numpy as np


def order_rtol(counters, directions=0):
    """bound on the relative difference of a tally word between two orders of its float64 additions"""
    events = counters["scatterings"] + counters["surfaceHits"] + counters["exitsTop"] + counters["photons"]
    return max(events * max(directions, 1), 64) * 2.0 ** -53


def assert_same_sums(a, b, counters, directions=0, what=None):
    """a, b: raw tally blocks (or any arrays of sums of the same increments); counters: the work counters of either run"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    rtol = order_rtol(counters, directions)
    # (radiance words may hold contributions of both signs -- truncated Legendre series go negative --: the bound is on the sum of
    # magnitudes, which no word exceeds by more than the largest word of the block)
    scale = float(np.abs(a).max()) if a.size else 0.0
    bad = np.abs(a - b) > rtol * (np.abs(a) + scale * 1e-3) + 1e-300
    assert not bad.any(), (what, int(bad.sum()), "first at", int(np.argmax(bad)), float(a.ravel()[np.argmax(bad)]), float(b.ravel()[np.argmax(bad)]),
                           "rtol", rtol, "largest relative difference", float((np.abs(a - b) / np.maximum(np.abs(a), 1e-300)).max()))


# normalised output fields are float32: the same float64 sums round to the same float32 value, or -- a sum sitting on a rounding
# boundary -- to its neighbour
F32_ULP = 2.0 ** -23
