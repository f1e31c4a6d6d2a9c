"""Time one configuration with several builds of the library: python3 tools/lib_compare.py <case> <photons> lib1.so lib2.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import i3rc_monte_carlo_model_amd as M
    M.build.LIB = sys.argv[4]
    M.build.needs_build = lambda: False
    sys.argv = [sys.argv[0], sys.argv[2], sys.argv[3]]
    exec(open(os.path.join(ROOT, "tools", "run_case.py")).read())
else:
    case, n = sys.argv[1], sys.argv[2]
    for lib in sys.argv[3:]:
        print(os.path.basename(lib), end=": ", flush=True)
        subprocess.call([sys.executable, __file__, "--one", case, n, os.path.abspath(lib)])
