#!/usr/bin/env python3
"""Extract numeric circuit/MLP constants (DATA, not code) from the reference tree.

Runs only in the build container (needs /root/reference).  Emits
`data/ow_gen_data.h`, a plain C header of `static const double` tables that both
the CPU oracle (oracle/) and the HIP library (openwurli_amd/csrc/) include.

Sources (reference file:line of each literal block):
  crates/openwurli-dsp/src/gen_tremolo.rs:77-1132,1829-1845  Twin-T oscillator MNA data
  crates/openwurli-dsp/src/mlp_weights.rs:12-603             2->16->16->11 MLP weights
  crates/openwurli-dsp/src/gen_preamp.rs (consts)             12-node preamp MNA data

Only array/scalar *literals* are carried over; every algorithm that consumes them
is restated by hand in oracle/ and csrc/.
"""
import re
import sys
from pathlib import Path

REF = Path("/root/reference/crates/openwurli-dsp/src")
OUT = Path(__file__).resolve().parent.parent / "data" / "ow_gen_data.h"

ARRAY_RE = re.compile(
    r"(?:pub(?:\(crate\))?\s+)?const\s+([A-Z0-9_]+)\s*:\s*(\[.*?\])\s*=\s*(\[.*?\]);", re.S)
SCALAR_RE = re.compile(
    r"(?:pub\s+)?const\s+([A-Z0-9_]+)\s*:\s*(f64|usize)\s*=\s*([^;]+);")


def parse_shape(ty: str):
    """'[[f64; N]; M]' -> ['M','N'] (outer first)."""
    dims = []
    t = ty.strip()
    while t.startswith("["):
        inner, _, dim = t[1:-1].rpartition(";")
        dims.append(dim.strip())
        t = inner.strip()
    return dims


def parse_numbers(body: str):
    body = re.sub(r"//[^\n]*", "", body)
    toks = re.findall(r"[-+]?(?:\d[\d_]*\.?[\d_]*(?:[eE][-+]?\d+)?|f64::INFINITY)", body)
    vals = []
    for t in toks:
        t = t.replace("_", "")
        vals.append("INFINITY" if "INFINITY" in t else t)
    return vals


def extract(path: Path, prefix: str, want_arrays, want_scalars, dim_env):
    src = path.read_text()
    out = []
    found = set()
    for m in ARRAY_RE.finditer(src):
        name, ty, body = m.group(1), m.group(2), m.group(3)
        if name not in want_arrays:
            continue
        dims = [int(dim_env.get(d, d)) for d in parse_shape(ty)]
        vals = parse_numbers(body)
        n = 1
        for d in dims:
            n *= d
        if len(vals) != n:
            raise SystemExit(f"{path.name}:{name}: expected {n} values, got {len(vals)}")
        cdims = "".join(f"[{d}]" for d in dims)
        out.append(f"OW_DATA_DECL double {prefix}{name}{cdims} = {{")
        row = dims[-1]
        for i in range(0, n, row):
            out.append("    " + ", ".join(vals[i:i + row]) + ",")
        out.append("};")
        found.add(name)
    for m in SCALAR_RE.finditer(src):
        name, ty, val = m.group(1), m.group(2), m.group(3).strip()
        if name not in want_scalars:
            continue
        val = val.replace("_", "")
        if "INFINITY" in val:
            continue
        if ty == "usize":
            out.append(f"#define {prefix}{name} {int(val)}")
        else:
            out.append(f"OW_DATA_DECL double {prefix}{name} = {val};")
        found.add(name)
    missing = (set(want_arrays) | set(want_scalars)) - found
    if missing:
        raise SystemExit(f"{path.name}: missing {sorted(missing)}")
    return out


def power_amp_tables():
    """melange 7-BJT Class-AB power amp (gen_power_amp.rs): N = 20 unknowns (18 nodes + 2 source rows), M = 16 ports (8 BJTs).
    Besides the literal tables, the EMITTED SPARSITY of process_sample is data too: which A_neg entries build_rhs touches
    (gen_power_amp.rs:8854-8913) and which two node voltages form each port voltage (:8936-8951)."""
    path = REF / "gen_power_amp.rs"
    src = path.read_text()
    arrays = ["G", "C", "A_NEG_DEFAULT", "A_NEG_BE_DEFAULT", "N_V", "N_I", "S_DEFAULT", "K_DEFAULT", "S_NI_DEFAULT",
              "S_BE_DEFAULT", "K_BE_DEFAULT", "S_NI_BE_DEFAULT", "RHS_CONST", "RHS_CONST_BE", "DC_OP", "DC_NL_I"]
    scalars = ["SAMPLE_RATE", "INPUT_RESISTANCE", "MAX_ITER", "DC_BLOCK_R"]
    out = ["", "/* ---- melange 7-BJT Class-AB power amp (gen_power_amp.rs), N=20 unknowns, M=16 NL ports ---- */"]
    out += extract(path, "PA_", arrays, scalars, {"N": 20, "M": 16, "NUM_OUTPUTS": 1})
    fields = ["IS", "VT", "BETA_F", "BETA_R", "NF", "NR", "ISE", "NE", "ISC", "NC", "SIGN", "VAF", "VAR", "IKF", "IKR", "VCRIT", "RB", "RC", "RE"]
    for f in fields:
        vals = []
        for d in range(8):
            m = re.search(rf"const DEVICE_{d}_{f}: f64 = ([^;]+);", src)
            if not m:
                raise SystemExit(f"gen_power_amp.rs: DEVICE_{d}_{f} missing")
            vals.append(m.group(1).strip().replace("_", ""))
        out.append(f"OW_DATA_DECL double PA_DEV_{f}[8] = {{" + ", ".join(vals) + "};")
    gp = [re.search(rf"const DEVICE_{d}_USE_GP: bool = (true|false);", src).group(1) for d in range(8)]
    out.append("OW_DATA_DECL double PA_DEV_USE_GP[8] = {" + ", ".join("1.0" if g == "true" else "0.0" for g in gp) + "};")
    body = src[src.index("pub fn process_sample("):]
    nz = re.findall(r"rhs\[(\d+)\] \+= state\.a_neg\[(\d+)\]\[(\d+)\] \* state\.v_prev\[(\d+)\];", body)
    assert nz and all(a == b and c == d for a, b, c, d in nz)
    out.append(f"#define PA_RHS_NNZ {len(nz)}")
    out.append("OW_DATA_DECL double PA_RHS_NZ_ROW[PA_RHS_NNZ] = {" + ", ".join(a for a, _, _, _ in nz) + "};")
    out.append("OW_DATA_DECL double PA_RHS_NZ_COL[PA_RHS_NNZ] = {" + ", ".join(c for _, _, c, _ in nz) + "};")
    pv = re.findall(r"p\[(\d+)\] = N_V\[(\d+)\]\[(\d+)\] \* v_pred\[(\d+)\] \+ N_V\[(\d+)\]\[(\d+)\] \* v_pred\[(\d+)\];", body)
    assert len(pv) == 16 and all(int(x[0]) == i for i, x in enumerate(pv))
    out.append("OW_DATA_DECL double PA_P_NODE_A[16] = {" + ", ".join(x[2] for x in pv) + "};")
    out.append("OW_DATA_DECL double PA_P_NODE_B[16] = {" + ", ".join(x[5] for x in pv) + "};")
    m = re.search(r"dc_block_x_prev: \[([-0-9.e+]+)\]", src)
    out.append(f"OW_DATA_DECL double PA_DC_BLOCK_X0 = {m.group(1)};")
    return out


def main():
    lines = [
        "/* GENERATED by tools/extract_constants.py -- numeric DATA only.",
        " * Circuit matrices and MLP weights of the OpenWurli reference (GPL-3.0-or-later);",
        " * see the generator for the reference file:line of every table.",
        " * Include with OW_DATA_DECL undefined for plain host tables, or define OW_DATA_DECL",
        " * (e.g. `__device__ const`) and OW_GEN_DATA_GUARD to instantiate a second copy. */",
        "#ifndef OW_DATA_DECL",
        "#define OW_DATA_DECL static const",
        "#endif",
        "#ifndef OW_GEN_DATA_GUARD",
        "#define OW_GEN_DATA_GUARD OW_GEN_DATA_H",
        "#endif",
        "#if !defined(OW_GEN_DATA_H) || defined(OW_GEN_DATA_REINCLUDE)",
        "#define OW_GEN_DATA_H",
        "",
        "/* ---- Twin-T tremolo oscillator (gen_tremolo.rs), N=7 nodes, M=4 NL ports ---- */",
    ]
    trem_arrays = [
        "G", "C", "A_NEG_DEFAULT", "A_NEG_BE_DEFAULT", "N_V", "N_I",
        "S_DEFAULT", "K_DEFAULT", "S_NI_DEFAULT",
        "S_BE_DEFAULT", "K_BE_DEFAULT", "S_NI_BE_DEFAULT",
        "RHS_CONST", "RHS_CONST_BE", "DC_OP", "DC_NL_I",
    ]
    trem_scalars = [
        "SAMPLE_RATE", "INPUT_RESISTANCE", "MAX_ITER",
        "DEVICE_0_IS", "DEVICE_0_VT", "DEVICE_0_BETA_F", "DEVICE_0_BETA_R",
        "DEVICE_0_NF", "DEVICE_0_NR", "DEVICE_0_VCRIT",
        "DEVICE_1_IS", "DEVICE_1_VT", "DEVICE_1_BETA_F", "DEVICE_1_BETA_R",
        "DEVICE_1_NF", "DEVICE_1_NR", "DEVICE_1_VCRIT",
    ]
    lines += extract(REF / "gen_tremolo.rs", "TREM_", trem_arrays, trem_scalars,
                     {"N": 7, "M": 4, "NUM_OUTPUTS": 1})
    lines += ["", "/* ---- note-on MLP 2->16->16->11 (mlp_weights.rs) ---- */"]
    mlp_arrays = ["W1", "B1", "W2", "B2", "W3", "B3", "TARGET_MEANS", "TARGET_STDS"]
    lines += extract(REF / "mlp_weights.rs", "MLP_", mlp_arrays, ["HIDDEN_SIZE"], {})
    lines += ["", "/* ---- melange 12-node DK preamp (gen_preamp.rs), N=12 nodes, M=3 NL ports ---- */"]
    pre_arrays = ["G", "C", "S_DEFAULT", "A_NEG_DEFAULT", "RHS_CONST", "K_DEFAULT", "N_V", "N_I", "S_NI_DEFAULT",
                  "S_BE_DEFAULT", "K_BE_DEFAULT", "S_NI_BE_DEFAULT", "A_NEG_BE_DEFAULT", "RHS_CONST_BE", "DC_OP", "DC_NL_I",
                  # thermal-noise stamps (node indices are emitted as doubles like every other table; they are small exact integers)
                  "NOISE_THERMAL_NODE_I", "NOISE_THERMAL_NODE_J", "NOISE_THERMAL_SQRT_INV_R_DEFAULT"]
    pre_scalars = ["SAMPLE_RATE", "INPUT_RESISTANCE", "POT_0_G_NOM",
                   "DEVICE_0_IS", "DEVICE_0_N_VT", "DEVICE_0_VCRIT",
                   "DEVICE_1_IS", "DEVICE_1_VT", "DEVICE_1_NF", "DEVICE_1_VCRIT",
                   "DEVICE_2_IS", "DEVICE_2_VT", "DEVICE_2_NF", "DEVICE_2_VCRIT"]
    lines += extract(REF / "gen_preamp.rs", "PRE_", pre_arrays, pre_scalars, {"N": 12, "M": 3, "NUM_OUTPUTS": 1, "NOISE_THERMAL_N": 11})
    lines += power_amp_tables()
    lines += ["", "#endif /* OW_GEN_DATA_H */", ""]
    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_text("\n".join(lines))
    print(f"wrote {OUT} ({len(lines)} lines)")


if __name__ == "__main__":
    sys.exit(main())
