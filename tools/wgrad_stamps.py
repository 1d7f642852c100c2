"""Where a workgroup of the batched wide weight gradient spends its cycles: builds csrc/dx_gemm.hip with -DDX_WG_STAMPS (s_memtime stamps,
written to a buffer nothing else reads) into tools/ab/lib_wgstamps.so and runs tools/microbench_wgrad.py's eight jobs once.
Diagnostic only: the product library carries no stamps."""
import ctypes, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
lib = os.environ.get("DX_STAMP_LIB") or os.path.join(REPO, "tools", "ab", "lib_wgstamps.so")
if not os.path.exists(lib) or '--build' in sys.argv:
    subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'ablation_build.py'), 'wgstamps', 'dx_gemm.hip', '-DDX_WG_STAMPS',
                    *[a for a in sys.argv[1:] if a.startswith('-D')]], check=True)
if '--build' in sys.argv:
    sys.exit(0)
os.environ['DX_LIB_PATH'] = lib
import torch
from ubisoft_laforge_daft_exprt_amd import ops, _lib
import tools.microbench_wgrad as mb   # noqa: E402


def main():
    dll = _lib.lib()
    stamps = torch.zeros(4096, 64, dtype=torch.int64, device='cuda')
    ctypes.CDLL(lib).dx_wgrad_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    mb.main()
    st = stamps.cpu().view(-1, 8, 8)
    st = st[st[:, 0, 0] > 0]
    f = lambda v: f'median {v.float().median():9.0f}  min {v.min():9d}  max {v.max():9d}'
    print(f'{st.shape[0]} workgroups stamped (s_memtime: shader cycles)')
    for w in range(8):
        s_ = st[:, w]
        n = s_[:, 6].clamp(min=1).float()
        print(f'  wave {w}: prologue {(s_[:, 1] - s_[:, 0]).float().median():7.0f}  loop {(s_[:, 2] - s_[:, 1]).float().median():8.0f}  epilogue {(s_[:, 3] - s_[:, 2]).float().median():7.0f}'
              f'  chunks {s_[:, 6].float().median():3.0f}  per chunk {((s_[:, 2] - s_[:, 1]).float() / n).median():6.0f}: load wait {(s_[:, 4].float() / n).median():6.0f}  barrier wait {(s_[:, 5].float() / n).median():6.0f}')
    print(f'  first start -> last end: {(st[:, :, 3].max() - st[:, :, 0].min()).item()} cycles')


if __name__ == '__main__':
    main()
