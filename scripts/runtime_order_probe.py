"""Which HIP runtime(s) end up in the process, by load order (development aid)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = {
    "engine_first": "from mind_the_gaps_amd.engine import Engine; e = Engine(0); import torch; print('torch sees', torch.cuda.is_available(), torch.cuda.device_count())",
    "torch_first": "import torch; print('torch sees', torch.cuda.is_available()); torch.zeros(1, device='cuda'); from mind_the_gaps_amd.engine import Engine; e = Engine(0); print('engine ok')",
    "torch_import_only_first": "import torch; from mind_the_gaps_amd.engine import Engine; e = Engine(0); print('torch sees', torch.cuda.is_available()); torch.zeros(1, device='cuda'); print('ok')",
}
TAIL = "\nimport re\nlibs = sorted({l.split()[-1] for l in open('/proc/self/maps') if re.search(r'amdhip64|hsa-runtime|hipfft|rocfft', l)})\nprint('\\n'.join(libs))"
for name, code in CODE.items():
    print("==", name, flush=True)
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\ntry:\n    %s\nexcept Exception as ex:\n    print('FAILED', type(ex).__name__, str(ex)[:200])%s" % (ROOT, code.replace("; ", "\n    "), TAIL)],
                       capture_output=True, text=True)
    print(r.stdout[-1500:], r.stderr[-600:], flush=True)
