"""Long-run parity soak: N engines play random parts for many seconds, every block compared with one CPU oracle engine each (output
within the parity bar, voice counts equal).
Usage: python tools/soak_parity.py [seconds] [engines] [preamp_kind] [power_amp_kind] [tremolo_kind] [--seed S] [--seeds S0,S1,...] [--ulp]
  (kinds as in include/openwurli_hip.h)
  --seed / --seeds   seed(s) of the random script (default 99, the script of rounds 3-5); several seeds print one line each and a summary
  --ulp              no GPU: the oracle against ITS OWN one-ulp build (the preamp's exp() off by one ulp) on the same script(s) -- what the
                     reference algorithm itself does under a different libm, the yardstick for the GPU's ratio
  --no-stop          do not stop at the first block outside the bar: the worst ratio over the whole length is what is wanted
  --ulp-voice        the same with the VOICE path's libm off by one ulp (cos / sin / exp behind every mode's rotation and decay)
  --ulp-r=K          the same with the plain build and its tremolo's r_ldr moved by K doubles (another libm's sin / pow / exp behind the CdS law)
Exit codes: 0 ran its length inside the bar; 1 mismatch; 3 (melange power amp only) ended at a divergence-guard event only one side took.

Absolute floor: ABS_FLOOR_SOAK = 2e-8 (tests/oracle_binding.py; DESIGN.md section 2 has its row; --floor X overrides).  The suites' four-note
scenarios use 2e-9 and their dense ones 5e-9; over minutes of dense play (up to 64 voices, volume up to 0.85, tremolo depth 0 .. 1) the
reference algorithm itself moves quiet samples by up to 3.4e-8 under a one-ulp exp(): the Newton stop of the preamp (|f| < 1e-9 V) is an
absolute threshold, and what reaches the output scales with volume^2 and the loop gain the LDR leaves (largest with the cell dark)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class Parted(Exception):
    """melange power amp: the soak ended at a guard event only one side took (the reference's documented instability)"""


def soak(seconds=60.0, n=4, pk=0, pak=0, tk=0, seed=99, ulp=False, verbose=True, switches=None, stop_on_mismatch=True, only=None, floor=None):
    """Returns dict(worst, branch, block, engine, blocks, wall).  Raises AssertionError on a mismatch, Parted for the melange power amp's
    unshared guard event.  only (ulp modes): the engines that are really rendered and compared -- the script draws its random numbers for all
    n, so engine k plays the same part as in the full run at 1 / n of the cost."""
    import oracle_binding as ob
    if floor is None:
        floor = max(ob.ABS_FLOOR_SOAK, ob.ABS_FLOOR_MELANGE_LIT_OUTPUT) if pk else (ob.ABS_FLOOR_SOAK_LFO if tk == 1 else ob.ABS_FLOOR_SOAK)
    sr, length = 48000.0, 512
    class _Phantom:         # an engine of the script that nobody listens to (only=...): every call is a no-op
        def __getattr__(self, name):
            return (lambda L: np.zeros(L, np.float32)) if name == "render" else (lambda *a, **k: 0)
    live = set(range(n)) if (only is None or not ulp) else set(only)
    cs = [ob.OracleEngine(sr, preamp_kind=pk, power_amp_kind=pak, tremolo_kind=tk) if k in live else _Phantom() for k in range(n)]
    if ulp:
        class _Pool:        # the oracle's one-ulp build behind the pool's interface
            def __init__(self):
                r_ulps = int(ulp.split(":")[1]) if isinstance(ulp, str) and ulp.startswith("r:") else 0      # "r:K": the plain build with its tremolo's R moved by K doubles
                self.e = [ob.OracleEngine(sr, perturbed=(False if r_ulps else ulp), preamp_kind=pk, power_amp_kind=pak, tremolo_kind=tk) if k in live else _Phantom() for k in range(n)]
                for k in live:
                    if r_ulps: self.e[k].set_r_ulp(r_ulps)
            def __getitem__(self, k): return self.e[k]
            def render(self, L): return np.stack([x.render(L) for x in self.e])
            def set_sample_rate(self, r):
                for x in self.e: x.set_sample_rate(r)
                for k in live:                                      # (a re-rating builds a new tremolo)
                    if isinstance(ulp, str) and ulp.startswith("r:"): self.e[k].set_r_ulp(int(ulp.split(":")[1]))
            def close(self):
                for x in self.e: x.close()
        g = _Pool()
    else:
        import openwurli_amd as ow
        g = ow.EnginePool(sr, n, preamp_kind=pk, power_amp_kind=pak, tremolo_kind=tk)
        for name, val in (switches or {}).items():      # probes: the same script under one of the pool's latched switches
            g.set_switch(name, val)
    g.set_sample_rate(sr)
    for c in cs:
        c.set_sample_rate(sr)
    rng = np.random.default_rng(seed)
    for k in range(n):
        for e in (g[k], cs[k]):
            e.set_tremolo_depth(0.25 * k); e.set_volume(0.35 + 0.1 * k); e.set_speaker_character(0.3 * (k % 3))
    held = [[] for _ in range(n)]
    blocks = int(seconds * sr / length)
    t0 = time.time()
    worst = dict(worst=0.0, branch="", block=-1, engine=-1)
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
            if k not in live:
                continue
            rep = ob.parity_report(go[k], c.render(length), abs_floor=floor)
            if rep["worst_ratio"] > worst["worst"]:
                worst = dict(worst=rep["worst_ratio"], branch=rep["worst_branch"], block=b, engine=k)
            voices_differ = (not ulp) and g[k].active_voice_count() != c.active_voice_count()
            if ((rep["n_bad"] and not ulp) or voices_differ) and (stop_on_mismatch or voices_differ):
                gr_now = (g[k].power_amp_diag().guard_resets, c.power_amp_diag()[3]) if pak else (0, 0)
                # Only a mismatch that BEGINS at an unshared guard event is the reference's own instability: the counts were equal after
                # the previous block (everything before was inside the bar) and differ after this one.  Anything else -- counts that were
                # already apart, or equal counts with different audio -- is a failure of the guard / reset / hold logic.
                if pak and guard_prev[k][0] == guard_prev[k][1] and gr_now[0] != gr_now[1]:
                    raise Parted("soak (power amp %d): GPU and oracle agreed (worst error / tolerance %.3f) for %.2f s; then engine %d took a guard reset "
                                 "on one side only (guard resets GPU %d, oracle %d) -- the reference's own one-ulp build parts the same way"
                                 % (pak, worst["worst"], b * length / sr, k, gr_now[0], gr_now[1]))
                raise AssertionError("MISMATCH at block %d engine %d %r voices %r guard resets (gpu, oracle) before / after %r %r"
                                     % (b, k, rep, (g[k].active_voice_count(), c.active_voice_count()), guard_prev[k], gr_now))
        if pak and not ulp:
            guard_prev = [(g[k].power_amp_diag().guard_resets, cs[k].power_amp_diag()[3]) for k in range(n)]
    worst.update(blocks=blocks, wall=time.time() - t0, seed=seed, floor=floor)
    if pak and not ulp:
        worst["guard_resets"] = [g[k].power_amp_diag().guard_resets for k in range(n)]
    g.close()
    for c in cs:
        c.close()
    if verbose:
        print("soak %s (preamp %d, power amp %d, tremolo %d, seed %d): %.0f s x %d engines, %d blocks, worst error / tolerance %.3f (the worst sample's "
              "tolerance was the %s term; block %d, engine %d), %.0f s wall%s"
              % (("of the oracle's one-ulp build" + (" (voice path)" if ulp == "voice" else "")) if ulp else "ok", pk, pak, tk, seed, seconds, n, blocks, worst["worst"], worst["branch"], worst["block"],
                 worst["engine"], worst["wall"], ("; power-amp guard resets per engine: %r" % worst["guard_resets"]) if "guard_resets" in worst else ""), flush=True)
    return worst


def main():
    argv = [a for a in sys.argv[1:]]
    seeds, ulp = [99], False
    if "--ulp" in argv:
        ulp = True; argv.remove("--ulp")
    if "--ulp-voice" in argv:
        ulp = "voice"; argv.remove("--ulp-voice")
    for a in list(argv):
        if a.startswith("--ulp-r="):          # the oracle against itself with the tremolo's R moved by K doubles (another libm's pow / exp / sin behind the CdS law)
            ulp = "r:" + a.split("=")[1]; argv.remove(a)
    floor = None
    if "--floor" in argv:
        i = argv.index("--floor"); floor = float(argv[i + 1]); del argv[i:i + 2]
    stop = True
    if "--no-stop" in argv:           # run the length whatever the bar says (voice counts still have to agree): the worst ratio per seed on record
        stop = False; argv.remove("--no-stop")
    for flag in ("--seed", "--seeds"):
        if flag in argv:
            i = argv.index(flag)
            seeds = [int(x) for x in argv[i + 1].split(",")]
            del argv[i:i + 2]
    seconds = float(argv[0]) if len(argv) > 0 else 60.0
    n = int(argv[1]) if len(argv) > 1 else 4
    pk = int(argv[2]) if len(argv) > 2 else 0
    pak = int(argv[3]) if len(argv) > 3 else 0
    tk = int(argv[4]) if len(argv) > 4 else 0
    rows = []
    for s in seeds:
        try:
            rows.append(soak(seconds, n, pk, pak, tk, seed=s, ulp=ulp, stop_on_mismatch=stop, floor=floor))
        except Parted as ex:
            print(ex)
            sys.exit(3)
        except AssertionError as ex:
            print(ex)
            sys.exit(1)
    if len(rows) > 1:
        print("summary (%s): seeds %r worst error / tolerance %r max %.3f" % (("one-ulp build" + (" (voice path)" if ulp == "voice" else "")) if ulp else "GPU", [r["seed"] for r in rows],
                                                                                    [round(r["worst"], 3) for r in rows], max(r["worst"] for r in rows)))


if __name__ == "__main__":
    main()
