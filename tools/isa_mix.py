#!/usr/bin/env python3
"""Instruction mix of a kernel's innermost loop, read off the shipped code object, priced with the measured issue costs of
profiles/r03_issue_calibration.txt (cycles a wave64 instruction occupies the float32 datapath of its SIMD: multiply-add class /
DPP / conversions / packed 4, two-operand and move class 2, transcendentals 8, v_mfma_f32_4x4x1 8, v_mfma_f32_16x16x4_f32 32;
binary16 matrix products run on the separate matrix pipe: 16 cycles each, listed beside).

    python tools/isa_mix.py [diffsptk_amd/lib/libdiffsptk_amd.so] [kernel-name-substring ...]      -> JSON per kernel

The loop taken is the LAST innermost loop of the kernel (for the mel-cepstral kernels: the Newton step; one pass = 16 frames)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TWO = {"v_mov_b32", "v_add_f32", "v_mul_f32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_f32", "v_sub_u32", "v_lshlrev_b32",
       "v_lshrrev_b32", "v_xor_b32", "v_subrev_f32", "v_subrev_u32", "v_max_f32", "v_min_f32", "v_cndmask_b32", "v_accvgpr_write_b32",
       "v_accvgpr_read_b32", "v_mov_b64"}
EIGHT = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_f32_dpp"}


def disassemble(lib):
    """{kernel symbol: [(address, mnemonic, text)]} of the gfx950 code objects bundled in `lib` (the .hip_fatbin section holds
    one clang offload bundle per translation unit: header magic, entry table (offset, size, triple), code objects)."""
    import struct

    tmp = "/tmp/isa_mix_%d" % os.getpid()
    os.makedirs(tmp, exist_ok=True)
    fat = os.path.join(tmp, "fat.bin")
    # llvm-objcopy without an output operand REWRITES its input: work on a private copy and send the output to a scratch file,
    # the product library is only ever read (tests/test_host_cpu.py checks its bytes before / after)
    import shutil

    priv = os.path.join(tmp, "lib_copy.so")
    shutil.copyfile(lib, priv)
    subprocess.run([OBJDUMP.replace("objdump", "objcopy"), "--dump-section", ".hip_fatbin=" + fat, priv, os.path.join(tmp, "lib_out.so")],
                   check=True)
    for f_ in (priv, os.path.join(tmp, "lib_out.so")):
        if os.path.exists(f_):
            os.remove(f_)
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    kernels, pos, n = {}, 0, 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        nb = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(nb):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off: off + tl].decode()
            off += tl
            if "gfx950" in triple and sz:
                co = os.path.join(tmp, "co%d.o" % n)
                n += 1
                open(co, "wb").write(data[i + o: i + o + sz])
                txt = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True).stdout
                cur = None
                for line in txt.splitlines():
                    m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                    if m:
                        cur = m.group(1)
                        kernels[cur] = []
                        continue
                    m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                    if m and cur:
                        kernels[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
                os.remove(co)
        pos = i + 24
    os.remove(fat)
    return kernels


def crossed_packed_f32(kernels):
    """{kernel: [instruction text]} of every packed float32 instruction with a SET op_sel bit -- a LOW result half that reads a
    HIGH source half -- in `kernels` (the output of disassemble()).  That is the instruction class of every transient wrong
    result DESIGN.md 4 recorded; the product must contain none (tests/test_host_cpu.py::test_no_crossed_packed_float32).
    `python tools/isa_mix.py --audit [lib.so]` prints the per-kernel counts."""
    bad = {}
    for name, ins in kernels.items():
        for _, op, txt in ins:
            if re.match(r"v_pk_\w+_f32", op):
                m = re.search(r"op_sel:\[([01,]+)\]", txt)
                if m and "1" in m.group(1):
                    bad.setdefault(name, []).append(op + " " + txt)
    return bad


def innermost_last_loop(ins):
    """(start, end) indices of the last backward branch's span that contains no other backward branch target span."""
    addr = {a: i for i, (a, _, _) in enumerate(ins)}
    loops = []
    for i, (a, op, txt) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            try:
                simm = int(txt.split()[0], 0)
            except (ValueError, IndexError):
                continue
            if simm >= 0x8000:
                simm -= 0x10000
            t = a + 4 + 4 * simm   # branch target = address of the next instruction + 4 * simm16
            if t in addr and addr[t] < i:
                loops.append((addr[t], i))
    inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
    return max(inner, key=lambda l: l[1] - l[0]) if inner else None


def mix(ins):
    c, cyc = collections.Counter(), collections.Counter()
    for _, op, txt in ins:
        base = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", op)
        if op.startswith("v_mfma_f32_4x4x1"):
            k, cy = "mfma_f32_4x4x1", 8
        elif op.startswith("v_mfma_f32_16x16x4"):
            k, cy = "mfma_f32_16x16x4", 32
        elif op.startswith("v_mfma"):
            k, cy = "mfma_f16_xdl", 0
        elif op.startswith("v_"):
            if base in EIGHT or op in EIGHT:
                k, cy = "valu_transcendental", 8
            elif base in TWO and not op.endswith("_dpp"):
                k, cy = "valu_two_operand", 2
            else:
                k, cy = "valu_multiply_add_class", 4
        elif op.startswith("ds_"):
            k, cy = "lds", 0
        elif op == "s_nop":
            k, cy = "s_nop", 0
        elif op.startswith("s_"):
            k, cy = "salu", 0
        else:
            k, cy = "vmem", 0
        c[k] += 1
        cyc[k] += cy
    return c, cyc


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(ROOT, "diffsptk_amd", "lib", "libdiffsptk_amd.so")
    names = [a for a in sys.argv[1:] if not a.endswith(".so")] or ["mcep_mfma_fwd_kernel_h", "mcep_mfma_bwd_kernel_h"]
    ks = disassemble(lib)
    res = {}
    for sym, ins in ks.items():
        if not any(n in sym for n in names) or not ins:
            continue
        lp = innermost_last_loop(ins)
        if lp is None:
            continue
        c, cyc = mix(ins[lp[0]: lp[1] + 1])
        vec = c["valu_multiply_add_class"] + c["valu_two_operand"] + c["valu_transcendental"]
        res[sym] = {"loop_instructions": lp[1] - lp[0] + 1, "counts": dict(c), "f32_datapath_cycles_per_pass": sum(cyc.values()),
                    "vector_instructions_per_pass": vec, "matrix_f32_instructions_per_pass": c["mfma_f32_4x4x1"] + c["mfma_f32_16x16x4"],
                    "xdl_cycles_per_pass": 16 * c["mfma_f16_xdl"], "frames_per_pass": 16}
    print(json.dumps(res, indent=1))


if __name__ == "__main__" and "--audit" in sys.argv:
    args = [a for a in sys.argv[1:] if a != "--audit"]
    ks = disassemble(args[0] if args else os.path.join(ROOT, "diffsptk_amd", "lib", "libdiffsptk_amd.so"))
    bad = crossed_packed_f32(ks)
    npk = sum(1 for ins in ks.values() for _, op, _t in ins if re.match(r"v_pk_\w+_f32", op))
    for k_, v_ in sorted(bad.items()):
        print("%5d  %s" % (len(v_), k_))
    print("%d kernels, %d packed float32 instructions, %d with a set op_sel bit in %d kernels" % (len(ks), npk, sum(map(len, bad.values())), len(bad)))
    sys.exit(1 if bad else 0)
elif __name__ == "__main__":
    main()
