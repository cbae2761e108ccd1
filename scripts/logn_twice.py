import sys, os, importlib.util
sys.path.insert(0, os.getcwd())
spec = importlib.util.spec_from_file_location("config3_probe", "scripts/config3_probe.py")
probe = importlib.util.module_from_spec(spec); spec.loader.exec_module(probe)
for k in range(3):
    d = probe.run(nsims=250, pdf="Lognormal")
    print("run %d: whole %.2f s, simulate %.3f s, refits %.2f s, p %.6f" % (k, d["whole_test_s"], d["seconds"]["simulate"], d["seconds"]["refit_null"] + d["seconds"]["refit_alt"], d["p_value"]), flush=True)
