for kc in 0 512 256 128; do echo "kc=$kc"; PD_DEBUG_SET="attn_bwd_kc=$kc" python tools/bench_attention.py 2>&1 | grep Lk | cut -c1-120; done
