#!/usr/bin/env python
"""Generator of the instruction-stream microbenchmarks behind the design of k_align2.

    python tools/gen_ubench.py ifetch  > tools/ubench_ifetch.hip     VOP2 issue rate vs. loop body size
    python tools/gen_ubench.py dstdep  > tools/ubench_dstdep.hip     issue rate vs. distance between writes of one register
    python tools/gen_ubench.py mix     > tools/ubench_mix.hip        mixes of 2-cycle and 4-cycle instruction forms
    python tools/gen_ubench.py replay [--lib platypus_amd/lib/libplatgpu.so] > tools/ubench_replay.hip
                                       the VALU instructions of k_align2's own inner loop (disassembled from the
                                       built library), alone and with one instruction of a slow form added

Each output is one self-contained HIP program (hipcc --offload-arch=gfx950 -O3 -o tools/ubench_X tools/ubench_X.hip)
that prints one line per kernel.  The outputs are not tracked: they are tens of thousands of lines of inline
assembly; results of the runs that shaped the kernel are under profiles/.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

CLOBBER = ", ".join('"v%d"' % i for i in range(127)) + ', "vcc"'
HEAD = "#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n\n"


def kernel(name, lines):
    body = "".join('        "%s\\n\\t"\n' % ln for ln in lines)
    return ("__global__ __launch_bounds__(256) void k_%s(uint32_t* out, int iters) {\n"
            "    for (int it = 0; it < iters; ++it) {\n        asm volatile(\n%s        ::: %s);\n    }\n"
            "    out[blockIdx.x * 256 + threadIdx.x] = iters;\n}\n\n" % (name, body, CLOBBER))


def main_fn(entries, fmt, args, waves=(1, 2), iters=400, extra=""):
    """entries: (label, kernel name, instructions per loop, extra initialisers)"""
    es = ", ".join('{"%s", k_%s, %d%s}' % e for e in entries)
    return ("struct E { const char* n; void (*f)(uint32_t*, int); int ni; %s};\n"
            "int main() {\n    E es[] = {%s};\n"
            "    uint32_t* out; if (hipMalloc(&out, 256*8*256*4) != hipSuccess) return 1;\n"
            "    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);\n"
            "    for (int bpc : {%s}) for (auto& e : es) {\n"
            "        const int blocks = 256 * bpc, iters = %d;\n        float best = 1e30f;\n"
            "        for (int rep = 0; rep < 3; ++rep) {\n            (void)hipEventRecord(e0, 0);\n"
            "            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, out, iters);\n"
            "            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);\n"
            "            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;\n        }\n"
            "        double li = (double)blocks * 256 * iters * e.ni;\n"
            "        double lanes = li / (best * 1e-3) / 256 / 2.4e9, cyc = 4.0 * 64.0 / lanes / 4.0; (void)cyc; (void)lanes;\n"
            "        printf(%s, %s);\n    }\n    return 0;\n}\n" % (extra, es, ", ".join(map(str, waves)), iters, fmt, args))


def gen_ifetch():
    """bodies of 64 .. 8192 VOP2 instructions (4 bytes each): rotating registers, 8 additions then 8 minima"""
    out, entries = [HEAD], []
    for n in (64, 256, 512, 1024, 2048, 4096, 8192):
        lines = []
        for i in range(n):
            op = "v_add_u16" if (i // 8) % 2 == 0 else "v_min_u16"
            d = 32 + i % 64
            lines.append("%s v%d, v%d, v%d" % (op, d, 32 + (i + 5) % 64, 32 + (i + 10) % 64))
        out.append(kernel(str(n), lines))
        entries.append((str(n), str(n), n, ""))
    out.append(main_fn(entries, '"body %5d instr (%6d B) waves/SIMD %d: %7.3f ms %6.1f lanes/clk/CU @2.4GHz\\n"',
                       "e.ni, e.ni * 4, bpc, best, lanes", iters=200))
    return "".join(out)


def gen_dstdep():
    """256 instructions of one opcode whose destinations cycle through a ring of 1 .. 64 registers; sources
    are never written, so only write-after-write distance varies"""
    out, entries = [HEAD], []
    for op in ("v_min_u16", "v_add_u16", "v_add_u32", "v_xor_b32", "v_min_f16"):
        for ring in (1, 2, 3, 4, 8, 16, 64):
            lines = ["%s v%d, v%d, v%d" % (op, 32 + i % ring, 100 + i % 5, 106 + i % 7) for i in range(256)]
            name = "%s_%d" % (op, ring)
            out.append(kernel(name, lines))
            entries.append((op, name, 256, ", %d" % ring))
    out.append(main_fn(entries, '"%-10s dst ring %2d  waves/SIMD %d: %7.3f ms %6.1f lanes/clk/CU @2.4GHz\\n"',
                       "e.n, e.nd, bpc, best, lanes", extra="int nd; "))
    return "".join(out)


MIXES = {   # name: (pattern of one group, ideal cycles per instruction if forms did not interact)
    "pkmin5_add4_xor1": (["v_pk_min_u16"] * 5 + ["v_add_u32"] * 4 + ["v_xor_b32"], 3.0),
    "pkmin_only": (["v_pk_min_u16"] * 10, 4.0),
    "add_u32_only": (["v_add_u32"] * 10, 2.0),
    "min16x5_add16x4_xor1": (["v_min_u16"] * 5 + ["v_add_u16"] * 4 + ["v_xor_b32"], 2.0),
    "interleaved_pkmin_add": (["v_pk_min_u16", "v_add_u32"] * 5, 3.0),
    "pkmin5_addu16x4_xor1": (["v_pk_min_u16"] * 5 + ["v_add_u16"] * 4 + ["v_xor_b32"], 3.0),
    "pkmin_pkadd": (["v_pk_min_u16", "v_pk_add_u16"] * 5, 4.0),
}


def gen_mix():
    out, entries = [HEAD], []
    for name, (pat, ideal) in MIXES.items():
        lines = []
        for i in range(1000):
            lines.append("%s v%d, v%d, v%d" % (pat[i % len(pat)], 32 + i % 64, 32 + (i + 17) % 64, 32 + (i + 31) % 64))
        out.append(kernel(name, lines))
        entries.append((name, name, 1000, ", %.2f" % ideal))
    out.append(main_fn(entries, '"%-26s waves/SIMD %d: %7.3f ms  %5.2f cycles/instr (ideal %4.2f)\\n"',
                       "e.n, bpc, best, 2.4e9 * best * 1e-3 / ((double)iters * e.ni * bpc), (double)e.ideal",
                       extra="float ideal; "))
    return "".join(out)


SLOW_FORMS = {   # one instruction of each form is appended to the replayed loop body
    "v_lshl_add_u64": "v_lshl_add_u64 v[8:9], v[20:21], 0, 0",
    "v_and_b32_e32": "v_and_b32_e32 v114, 0xff00, v4",
    "v_mov_b32": "v_mov_b32_e32 v114, v4",
    "v_min_u16_sdwa": "v_min_u16_sdwa v114, v4, v5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1",
    "v_min3_u16": "v_min3_u16 v114, v4, v5, v6",
    "v_lshlrev_b32_sdwa": "v_lshlrev_b32_sdwa v114, v4, v5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1",
    "v_bfe_u32": "v_bfe_u32 v114, v4, 8, 8",
    "v_and_b32_sdwa": "v_and_b32_sdwa v114, v4, v5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0",
}


def inner_loop_valu(lib):
    """VALU instructions of the longest backward-branch loop of pg::k_align2 in the built library"""
    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as tmp:
        co, fat = os.path.join(tmp, "co"), os.path.join(tmp, "fat")
        subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib,
                               os.path.join(tmp, "unused")])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.DEVNULL)
        text = subprocess.check_output([os.path.join(llvm, "llvm-objdump"), "-d", "--no-show-raw-insn",
                                        "--symbolize-operands", co], text=True)
    m = re.search(r"<_ZN2pg8k_align2ENS_10Align2ArgsE>:\n(.*?)\n[0-9a-f]+ <_Z", text, re.S)
    if not m:
        raise SystemExit("k_align2 not found in %s" % lib)
    lines = [ln.strip() for ln in m.group(1).splitlines()]
    label_at = {}
    for i, ln in enumerate(lines):
        lm = re.match(r"[0-9a-f]+ <(L\d+)>:", ln)
        if lm:
            label_at[lm.group(1)] = i
    best = (0, 0, 0)
    for i, ln in enumerate(lines):
        bm = re.match(r"s_c?branch\w* (L\d+)", ln)
        if bm and bm.group(1) in label_at and label_at[bm.group(1)] < i:
            n = sum(1 for x in lines[label_at[bm.group(1)]:i] if x.startswith("v_"))
            if n > best[0]:
                best = (n, label_at[bm.group(1)], i)
    body = [re.sub(r"\s+//.*$", "", x) for x in lines[best[1]:best[2]] if x.startswith("v_")]
    # keep register-only arithmetic: no memory, no reads of SGPRs / exec (they would be undefined in the replay)
    return [x for x in body if not re.search(r"\bs\d+|\bs\[|exec|vcc|readlane|readfirstlane", x)]


def gen_replay(lib):
    core = inner_loop_valu(lib)
    out, entries = [HEAD, "// %d register-only VALU instructions of k_align2's inner loop\n" % len(core)], []
    out.append(kernel("core", core))
    entries.append(("core", "core", len(core), ""))
    for name, ins in SLOW_FORMS.items():
        out.append(kernel("core_plus_" + name, [ins] + core))
        entries.append(("core_plus_" + name, "core_plus_" + name, len(core) + 1, ""))
    out.append(main_fn(entries, '"%-34s %d VALU/loop  waves/SIMD %d: %7.3f ms %6.1f lanes/clk/CU @2.4GHz\\n"',
                       "e.n, e.ni, bpc, best, lanes", waves=(2,)))
    return "".join(out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["ifetch", "dstdep", "mix", "replay"])
    ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                   "platypus_amd", "lib", "libplatgpu.so"))
    a = ap.parse_args()
    sys.stdout.write({"ifetch": gen_ifetch, "dstdep": gen_dstdep, "mix": gen_mix,
                      "replay": lambda: gen_replay(a.lib)}[a.what]())
