"""Regenerates the table of recordable entry points inside partdistillation_amd/csrc/cmdbuf.hip from include/*.h: every `int pd_*(...)`
function except the host-only queries.  Run after adding a C-ABI function that a recorded region calls:  python tools/gen_cmdbuf_table.py"""
import glob, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = {"pd_abi_version", "pd_conv_bf16_supported", "pd_igemm_bf16_supported", "pd_gemm_wgrad_f16x2_takes_wide_tiles",
        "pd_point_sample_planar_bwd_needs_zero", "pd_point_sample_planar_bwd_needs_zero_n", "pd_msda_backward_last_gate", "pd_msda_fused_supported", "pd_gemm_tn_f16x2_which", "pd_debug_set",
        "pd_cmd_replay", "pd_cmd_fn_index", "pd_cmd_fn_nargs", "pd_igemm_bf16_time", "pd_mx8_gemm_supported"}
names = set()
for f in glob.glob(os.path.join(ROOT, "include", "*.h")):
    names |= set(re.findall(r"^int (pd_\w+)\(", open(f).read(), re.M))
names = sorted(names - SKIP)
p = os.path.join(ROOT, "partdistillation_amd", "csrc", "cmdbuf.hip")
s = open(p).read()
a, b = s.index("const Entry kTable[] = {\n") + len("const Entry kTable[] = {\n"), s.index("};\nconstexpr int kCount")
s = s[:a] + "".join(f"  PD_E({n}),\n" for n in names) + s[b:]
open(p, "w").write(s)
print(len(names), "recordable entry points")
