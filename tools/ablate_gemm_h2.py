"""Times the ablated copies of pd_gemm_tn_f16x2 built by tools/ablate_gemm_h2.sh (one child process per copy: the library is loaded
once per process).  Results are NOT valid products: the point is which part of the step the kernel's time is made of."""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
NAMES = {1: "no-mfma", 2: "no-ldsread", 4: "no-split/ldswrite", 8: "no-gload", 16: "no-cstore", 32: "no-scales", 64: "no-wload(rows)"}


def child(k):
    import torch
    from partdistillation_amd import lib
    lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"libpd_abl_{k}.so")
    L = lib.load()
    from partdistillation_amd.functions import gemm
    out = []
    for M, N, K, tile in [(43520, 256, 256, 61), (43520, 1024, 256, 61), (43520, 256, 256, 70), (43520, 256, 1024, 70), (43520, 1024, 256, 70)]:
        a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
        aa, wa = gemm.row_amax(a), gemm.row_amax(w)
        L.pd_debug_set(b"f16x2_tile", tile)
        f = lambda: gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): f()
        e1.record(); torch.cuda.synchronize()
        out.append(f"{e0.elapsed_time(e1) / 40 * 1e3:6.1f}")
    kk = int(str(k).split("x")[0])
    print(f"ABL {k:>4s} {'+'.join(v for b, v in NAMES.items() if kk & b) or 'full':45s} " + "  ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        print("columns (us): row stream 256<-256 | row stream 1024<-256 | tiled 256<-256 | tiled 256<-1024 | tiled 1024<-256 (tiled: the guarded step, which carries the ablation bits); M = 43520")
        for k in sys.argv[1:]:
            subprocess.run([sys.executable, __file__, "--child", str(k)], check=False)
