"""`preamp-bench alias-audit` on the device: the canonical three-note sweep gated against the v0.5.1 baseline, then all 64 keys x 8
velocities in one pool.  Exit code 1 if the reference's gate (tests/alias_audit_regression.rs:29-30) fails."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from openwurli_amd import alias_audit as aa
    base = json.load(open(os.path.join(ROOT, "tests", "golden", "alias_audit_v0_5_1.json")))
    t0 = time.perf_counter()
    sweep = aa.run_sweep()
    t1 = time.perf_counter()
    ok = True
    for ent, got in zip(base["entries"], sweep):
        r = got.result
        d_step, d_hf = r.max_step_up_db - ent["max_step_up_db"], r.hf_band_dbc - ent["hf_band_dbc"]
        good = d_step <= 1.5 and d_hf <= 2.0
        ok = ok and good
        print(f"note {got.note}: f0 {r.f0_hz:.4f} Hz  H1 {r.h1_dbfs:.3f} dBFS  step-up {r.max_step_up_db:+.3f} dB ({d_step:+.3f} vs v0.5.1)  "
              f"HF band {r.hf_band_dbc:.3f} dBc ({d_hf:+.3f})  {'ok' if good else 'REGRESSION'}")
    notes = [n for n in range(33, 97) for _ in range(8)]
    vels = [v for _ in range(33, 97) for v in (20, 35, 50, 65, 80, 95, 110, 127)]
    t2 = time.perf_counter()
    grid = aa.run_notes(notes, vels)
    t3 = time.perf_counter()
    worst = max(grid, key=lambda r: r.max_step_up_db)
    print(f"sweep {t1 - t0:.2f} s; 512-stimulus grid {t3 - t2:.2f} s; worst step-up on the grid {worst.max_step_up_db:+.2f} dB at f0 {worst.f0_hz:.1f} Hz")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
