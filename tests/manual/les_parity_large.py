"""The LES stratocumulus + Rayleigh domain of the reference's tool chain at a larger sample than the suite's test: GPU (1e7 photons in
40 batches) against the oracle (2.4e6 in 48 batches on the host's cores), hybrid phase functions, two radiance directions.
usage: python3 tests/manual/les_parity_large.py"""
import os, sys, tempfile
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np

MUS, PHIS = [1.0, 0.6], [0.0, 135.0]


def setup():
    import i3rc_monte_carlo_model_amd as M
    from tests.test_fortran_shell import _tool_chain_domain
    dom = M.read_Domain(_tool_chain_domain("les_stcu_rayleigh", tempfile.mkdtemp()))
    nz, ny, nx = dom.shape
    def full(c, key, dtype):
        a = np.zeros((nz, ny, nx), dtype); z0 = c["zbase"] - 1
        a[z0:z0 + c[key].shape[0]] = np.broadcast_to(c[key], (c[key].shape[0], ny, nx)); return a
    d = dict(xe=dom.x, ye=dom.y, ze=dom.z, ext=np.stack([full(c, "ext", np.float32) for c in dom.components]),
             ssa=np.stack([full(c, "ssa", np.float32) for c in dom.components]), pf=np.stack([full(c, "pfi", np.int32) for c in dom.components]))
    inv = [c["table"].inverse_table(10001) for c in dom.components]
    fwd = [c["table"].forward_table(10001) for c in dom.components]
    return dom, d, inv, fwd


def oracle_batches(first, count, n):
    from oracle import pyoracle as O
    dom, d, inv, fwd = setup()
    hyb = [O.hybrid_tables(f, 7.0) for f in fwd]
    o = O.Integrator(d["xe"], d["ye"], d["ze"], d["ext"], d["ssa"], d["pf"], inv, hyb, fwd)
    o.specify(intensityMus=MUS, intensityPhis=PHIS, useRRForIntensity=1, zetaMin=0.3, surfaceAlbedo=0.06, useHybrid=1, numOrdersOrig=0)
    out = []
    for b in range(first, first + count):
        rng = O.RandomNumberSequence([77, b]); r = o.compute(rng, *O.photons_directional(rng, 0.5, 0.0, n))
        out.append([float(r["fluxUp"].mean()), float(r["fluxDown"].mean()), float(r["fluxAbsorbed"].mean())] + [float(r["intensity"][k].mean()) for k in range(2)])
    return out


if __name__ == "__main__":
    import i3rc_monte_carlo_model_amd as M
    from oracle import pyoracle as O
    O.build()
    cores = min(16, len(os.sched_getaffinity(0)))
    with ProcessPoolExecutor(cores) as ex:
        futs = [ex.submit(oracle_batches, 1 + 3 * k, 3, 50000) for k in range(16)]
        dom, d, inv, fwd = setup()
        hyb = [O.hybrid_tables(f, 7.0) for f in fwd]
        g = M.new_Integrator(dom)
        g.specifyParameters(surfaceAlbedo=0.06, minInverseTableSize=10001, intensityMus=MUS, intensityPhis=PHIS, useRussianRouletteForIntensity=True, zetaMin=0.3,
                            useHybridPhaseFunsForIntenCalcs=True, hybridPhaseFunWidth=7.0, numOrdersOrigPhaseFunIntenCalcs=0)
        for k in range(2):
            g.set_tables(k + 1, inverse=inv[k], forward=hyb[k], forward_orig=fwd[k])
        gr = []
        for b in range(1, 41):
            r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((5, b)), M.new_PhotonStream(0.5, 0.0, 250000))
            gr.append([float(r["fluxUp"].mean()), float(r["fluxDown"].mean()), float(r["fluxAbsorbed"].mean())] + [float(r["intensity"][k].mean()) for k in range(2)])
        print("kernel", g.kernel_name())
        orr = [row for f in futs for row in f.result()]
    gr, orr = np.array(gr), np.array(orr)
    for k, name in enumerate(("fluxUp", "fluxDown", "fluxAbsorbed", "radiance mu 1.0", "radiance mu 0.6")):
        mg, mo = gr[:, k].mean(), orr[:, k].mean()
        se = np.sqrt(gr[:, k].var(ddof=1) / len(gr) + orr[:, k].var(ddof=1) / len(orr))
        print(f"{name:18s} GPU {mg:.6f}  oracle {mo:.6f}  difference {mg - mo:+.2e} = {abs(mg - mo) / se:.2f} combined standard errors of {se:.1e}")
