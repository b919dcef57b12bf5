#!/usr/bin/env python3
"""Diagnostic build with in-kernel phase stamps (never part of the product build).

Rewrites the `// [stamp:...]` markers of k_encoder_i8 in csrc/encoder.hip into s_memtime stamps (one asm statement
with its own lgkmcnt(0), fenced by sched_barrier, as cdna_hip_programming.md section 7 prescribes), adds a debug buffer
to the encoder handle, and prints per-tile cycle shares when the handle is destroyed (SMK_ENC_STAMPS=1).
    python tools/stamp_build.py apply   # patch + build        python tools/stamp_build.py restore
Stamp values go only to the debug buffer; never quote the run time of this build, only the shares.
"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "smokephysai_amd", "csrc")
FILES = ["encoder.hip", "encoder.h", "api.hip"]
NAMES = ["conv1", "bar_max", "quant", "bar_a1", "kloop", "epilogue", "bar_end"]

def apply():
    for f in FILES:
        shutil.copy(os.path.join(C, f), os.path.join("/tmp", f + ".prestamp"))
    s = open(os.path.join(C, "encoder.h")).read()
    s = s.replace("    unsigned short *w2q;", "    unsigned long long *stamps;\n    unsigned short *w2q;")
    open(os.path.join(C, "encoder.h"), "w").write(s)
    s = open(os.path.join(C, "api.hip")).read()
    s = s.replace("    enc->e.w2q = enc->blob16;", "    enc->e.stamps = nullptr; if (getenv(\"SMK_ENC_STAMPS\")) { (void)hipMalloc((void**)&enc->e.stamps, 8*8*4096); (void)hipMemset(enc->e.stamps, 0, 8*8*4096); }\n    enc->e.w2q = enc->blob16;")
    s = s.replace('#include <map>', '#include <stdlib.h>\n#include <map>')
    names = " ".join(f"{n}=%llu" for n in NAMES)
    args = ", ".join(f"tot[{i}]/tot[7]" for i in range(len(NAMES)))
    s = s.replace("int smk_encoder_destroy(smk_encoder *enc) {\n    if (!enc) return SMK_OK;\n    (void)hipSetDevice(enc->device);",
                  "int smk_encoder_destroy(smk_encoder *enc) {\n    if (!enc) return SMK_OK;\n    (void)hipSetDevice(enc->device);\n    if (enc->e.stamps) { static unsigned long long h[8*4096]; (void)hipDeviceSynchronize(); (void)hipMemcpy(h, enc->e.stamps, sizeof(h), hipMemcpyDeviceToHost); unsigned long long tot[8]={0}; int n=0; for (int w=0; w<4096; ++w) { if (!h[w*8+7]) continue; ++n; for (int k=0;k<8;++k) tot[k]+=h[w*8+k]; } if (n) fprintf(stderr, \"STAMPS waves=%d tiles=%llu per-tile cycles: " + names + "\\n\", n, tot[7], " + args + "); }")
    open(os.path.join(C, "api.hip"), "w").write(s)
    s = open(os.path.join(C, "encoder.hip")).read()
    stamp = 'unsigned long long {v}; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"({v}) :: "memory"); __builtin_amdgcn_sched_barrier(0);'
    s = s.replace("// [stamp:begin]", "unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};")
    for i in range(8):
        s = s.replace(f"// [stamp:T{i}]", stamp.format(v=f"T{i}"))
    s = s.replace("// [stamp:accumulate]", " ".join(f"acc_t[{i}] += T{i+1} - T{i};" for i in range(7)) + " acc_t[7] += 1;")
    s = s.replace("// [stamp:end]", "if (e.stamps && lane == 0) { const int slot = (blockIdx.x * 4 + wave) & 4095; for (int k = 0; k < 8; ++k) e.stamps[slot * 8 + k] = acc_t[k]; }")
    open(os.path.join(C, "encoder.hip"), "w").write(s)
    subprocess.check_call(["make", "-s", "-C", C])

def restore():
    for f in FILES:
        shutil.copy(os.path.join("/tmp", f + ".prestamp"), os.path.join(C, f))
    subprocess.check_call(["make", "-s", "-C", C])

if __name__ == "__main__":
    {"apply": apply, "restore": restore}[sys.argv[1]]()
