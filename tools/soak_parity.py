"""Long-run parity soak (not part of the suites): N engines play random parts for many seconds, every block compared with one CPU
oracle engine each (output within the parity bar, voice counts equal).
Usage: python tools/soak_parity.py [seconds] [engines] [preamp_kind] [power_amp_kind] [tremolo_kind]   (kinds as in include/openwurli_hip.h)
Exit codes: 0 ran its length inside the bar; 1 mismatch; 3 (melange power amp only) ended at a divergence-guard event only one side took.

Absolute floor: 5e-9.  The suites use 2e-9, four times what a one-ulp exp() perturbation moves the oracle in the 4-note scenario of
tests/test_oracle_sensitivity.py; under dense play (up to 64 voices, volume up to 0.65, tremolo depth up to 1) the same experiment --
oracle against its own perturbed build on THIS script -- moves quiet samples by up to 3.1e-9: the Newton stop of the preamp
(|f| < 1e-9 V) is an absolute threshold and what reaches the output scales with volume^2 and the tremolo's gain swing."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle_binding as ob
    import openwurli_amd as ow
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    pk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    pak = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    tk = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    # melange preamp: its own (literal-rebuild) floor; everything else: the dense-play floor explained above
    floor = max(ob.ABS_FLOOR_DENSE, ob.ABS_FLOOR_MELANGE_LIT_OUTPUT) if pk else ob.ABS_FLOOR_DENSE
    sr, length = 48000.0, 512
    g = ow.EnginePool(sr, n, preamp_kind=pk, power_amp_kind=pak, tremolo_kind=tk); g.set_sample_rate(sr)
    cs = [ob.OracleEngine(sr, preamp_kind=pk, power_amp_kind=pak, tremolo_kind=tk) for _ in range(n)]
    for c in cs:
        c.set_sample_rate(sr)
    rng = np.random.default_rng(99)
    for k in range(n):
        for e in (g[k], cs[k]):
            e.set_tremolo_depth(0.25 * k); e.set_volume(0.35 + 0.1 * k); e.set_speaker_character(0.3 * (k % 3))
    held = [[] for _ in range(n)]
    blocks = int(seconds * sr / length)
    worst, t0 = 0.0, time.time()
    guard_prev = [(0, 0)] * n
    for b in range(blocks):
        for k in range(n):
            if rng.random() < 0.08 + 0.03 * k:
                note, vel = int(rng.integers(33, 97)), float(rng.uniform(0.2, 1.0))
                for e in (g[k], cs[k]):
                    e.note_on(note, vel)
                held[k].append(note)
            if held[k] and rng.random() < 0.07:
                note = held[k].pop(int(rng.integers(0, len(held[k]))))
                for e in (g[k], cs[k]):
                    e.note_off(note)
            if rng.random() < 0.01:
                on = bool(rng.integers(0, 2))
                for e in (g[k], cs[k]):
                    e.set_sustain(on)
            if rng.random() < 0.002:
                d = float(rng.uniform(0.0, 1.0))
                for e in (g[k], cs[k]):
                    e.set_tremolo_depth(d)
        go = g.render(length)
        for k, c in enumerate(cs):
            rep = ob.parity_report(go[k], c.render(length), abs_floor=floor)
            worst = max(worst, rep["worst_ratio"])
            if rep["n_bad"] or g[k].active_voice_count() != c.active_voice_count():
                gr_now = (g[k].power_amp_diag().guard_resets, c.power_amp_diag()[3]) if pak else (0, 0)
                # Only a mismatch that BEGINS at an unshared guard event is the reference's own instability: the counts were equal after
                # the previous block (everything before was inside the bar) and differ after this one.  Anything else -- counts that were
                # already apart, or equal counts with different audio -- is a failure of the guard / reset / hold logic.
                if pak and guard_prev[k][0] == guard_prev[k][1] and gr_now[0] != gr_now[1]:
                    # The melange power amp's divergence guard is not stable against last-bit differences of its input (the oracle
                    # parts from its own one-ulp build the same way: tests/test_oracle_sensitivity.py); sample-for-sample parity
                    # of an engine with this amp ends at the first guard event the two sides do not share.  Not a failure.
                    print("soak (power amp %d): GPU and oracle agreed (worst error / tolerance %.3f) for %.2f s; then engine %d took a guard reset on "
                          "one side only (guard resets GPU %d, oracle %d) -- the reference's own one-ulp build parts the same way"
                          % (pak, worst, b * length / sr, k, g[k].power_amp_diag().guard_resets, c.power_amp_diag()[3]))
                    sys.exit(3)      # distinct from success: the soak ENDED here, it did not run its length
                print("MISMATCH at block", b, "engine", k, rep, g[k].active_voice_count(), c.active_voice_count(), "guard resets (gpu, oracle) before / after",
                      guard_prev[k], gr_now)
                sys.exit(1)
        if pak:
            guard_prev = [(g[k].power_amp_diag().guard_resets, cs[k].power_amp_diag()[3]) for k in range(n)]
    extra = ""
    if pak:
        extra = "; power-amp guard resets per engine: " + str([g[k].power_amp_diag().guard_resets for k in range(n)])
    print("soak ok (preamp %d, power amp %d, tremolo %d): %.0f s x %d engines, %d blocks, worst error / tolerance %.3f, %.0f s wall%s"
          % (pk, pak, tk, seconds, n, blocks, worst, time.time() - t0, extra))


if __name__ == "__main__":
    main()
