"""Does a workload's rate depend on where its buffers land?  A dummy device allocation of PAD bytes before the integrator is made
shifts every later allocation; one process per PAD.   usage: pad_sweep.py <workload> <photons> [pad bytes ...]"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 3 and sys.argv[3] == "--one":
    pad = int(sys.argv[4])
    hip = ctypes.CDLL("libamdhip64.so")
    p = ctypes.c_void_p()
    if pad:
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(pad)) == 0
    import i3rc_monte_carlo_model_amd as M
    from tools import workloads as W
    if os.environ.get("I3RC_LIB"):
        M.build.LIB = os.path.abspath(os.environ["I3RC_LIB"]); M.build.needs_build = lambda: False
    name, w = W.get(sys.argv[1]); n = int(float(sys.argv[2]))
    g, d = W.make_integrator(w)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(w["mu0"], 0.0, 1))
    out = []
    for k in range(2):
        g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1 + k)), M.new_PhotonStream(w["mu0"], 0.0, n))
        out.append(g.kernel_ms())
    print(f"pad {pad:>9d} (device pointer {p.value or 0:#x}): " + "  ".join(f"{n / t * 1e3:.3e} photons/s ({t:.1f} ms)" for t in out), flush=True)
else:
    pads = [int(x) for x in sys.argv[3:]] or [0, 4096, 65536, 1 << 20, (1 << 20) + 65536, 2 << 20, 3 << 20, 5 << 20, 8 << 20, 33 << 20]
    for pad in pads:
        subprocess.call([sys.executable, __file__, sys.argv[1], sys.argv[2], "--one", str(pad)])
