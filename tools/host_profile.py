"""Tool (not product): cProfile of the host side of a bench workload (where the per-step enqueue time goes)."""
import cProfile, pstats, sys, io
sys.path.insert(0, '.')
import bench
argv = sys.argv[1:] + ['--no-cpu-baseline', '--no-extra-legs']
bench.main(argv + ['--steps', '3', '--warmup', '3'], emit=False)       # warm caches / lazy builds
pr = cProfile.Profile()
pr.enable()
bench.main(argv + ['--steps', '20', '--warmup', '2'], emit=False)
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
s2 = io.StringIO()
pstats.Stats(pr, stream=s2).sort_stats('tottime').print_callers("method 'to' of")
print(s2.getvalue()[:5000])
