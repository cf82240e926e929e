#!/usr/bin/env python3
"""Prints nm_selftest_mfma_model for both matrix instructions of the matcher's screens (profiles/r04_*_mfma_model.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import niftymatch_amd as nm

for instr, name, screen in ((1, "v_mfma_f32_32x32x16_f16 (coarse pass of the two-stage screen)", 2),
                            (0, "v_mfma_f32_32x32x16_bf16 (bf16x3 screen, norm k-slots)", 1)):
    r = nm.selftest_mfma_model(instr, n_random=1 << 22, n_chains=1 << 15)
    b = nm.match_accum_budget(screen)
    print(name)
    for k, v in r.items():
        print("    %-28s %.6g" % (k, v))
    print("    %-28s %.6g   (chain_coeff / budget = %.3f, subnormal families %.3f)" %
          ("accumulation budget", b, r["chain_coeff"] / b, r["chain_coeff_subnormal"] / b))
