"""Which part of the GPU path carries a soak's worst error?  The script of tools/soak_parity.py (one seed, a few seconds) under the pool's
switches that exchange one formulation for another: the steady voice kernel and its variants (deviations 9, 13, 15) against the
general kernel (the reference's own formulation per voice: force_general), the row / quad / two-launch chains (bit-identical: a control).
Usage: python tools/probe_soak_variants.py [seconds] [engines] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak_parity

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for name, sw in (("default", {}), ("general voice kernel for every engine", {"force_general": 1}),
                 ("no attack / steal / release variants", {"voice_attack": 0, "voice_steal": 0, "voice_release": 0}),
                 ("two-launch chain (control: bit-identical)", {"chain_fused": 0})):
    r = soak_parity.soak(seconds, n, seed=seed, switches=sw, verbose=False, stop_on_mismatch=False)
    print("%-46s worst error / tolerance %.3f (%s term, block %d, engine %d)" % (name, r["worst"], r["branch"], r["block"], r["engine"]), flush=True)
