#!/usr/bin/env python3
"""tools/loop_microbench_gen.py -- generates tools/loop_microbench.hip: an inner loop of the FAST kernel as hipcc compiled it
(instruction for instruction, with the registers hipcc allocated) run in isolation under s_memtime, plus edited variants, to
see what an edited loop body would buy BEFORE touching the kernel.

    python3 tools/loop_microbench_gen.py [listing.s] > tools/loop_microbench.hip
    hipcc -O3 --offload-arch=gfx950 tools/loop_microbench.hip -o tools/loop_microbench

The default listing is tools/data/round2_lds_ring_kernel_f32_R2_S8.s: the round-2 kernel that staged the bodies j in a
per-wave LDS ring (git history: the commit before "FAST: bodies j through scalar loads"), hipcc's own output for
integrate_bodies_fast<float,2,8,2>.  The variants "body j as SGPR operands" of that loop are what motivated the present
kernel (65.4 -> 61.8 cycles per interaction pair).

Each kernel: 1024-thread workgroups (16 waves = 4 per SIMD), one per CU; every wave runs OUTER x 32 iterations of the
loop reading a 64 KiB LDS buffer; true cycles from s_memtime (shader clock), so DVFS does not blur the result.  Reports
cycles per packed interaction pair per SIMD.
"""
import re
import sys

import os

src = open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "round2_lds_ring_kernel_f32_R2_S8.s")).read().split("\n")
KERNEL = "_ZN2nb12_GLOBAL__N_121integrate_bodies_fastIfLi2ELi8ELi2EEEvNS_5ShardIT_EE"  # the production geometry at 262 144 bodies
start = next(i for i, l in enumerate(src) if l.startswith(KERNEL + ":"))
# every depth-2 inner loop of the kernel (round 2: a generic-mass loop and a unit-mass loop), loop control stripped:
# the harness supplies its own counter
loops = []
for i in range(start, len(src)):
    if src[i].startswith(".Lfunc_end"):
        break
    if "Inner Loop Header: Depth=2" in src[i]:
        label = src[i - 1].split(":")[0].strip()
        end = next(k for k in range(i, len(src)) if "s_cbranch" in src[k] and label in src[k])
        lines = [re.sub(r"\s*;.*$", "", l.strip()) for l in src[i + 1:end] if l.strip() and not l.strip().startswith(";")]
        lines = [l for l in lines if not (l.startswith("s_add_i32") or l.startswith("s_cmpk") or l.startswith("s_cmp_"))]
        loops.append((label, lines))
body = loops[0][1]

OUTER = 512
variants = []


def variant(name, lines, pairs_per_iter=8, prologue=()):
    variants.append((name, lines, pairs_per_iter, tuple(prologue)))


for label, lines in loops:
    variant(f"as compiled {label}: " + ", ".join(f"{sum(1 for l in lines if l.startswith(op))} {op}" for op in ("v_pk_", "v_rsq", "v_mov", "ds_read", "s_nop")), lines)

# no LDS reads, no waits: pure VALU stream
variant("VALU only (ds_read + s_waitcnt removed)", [l for l in body if not l.startswith("ds_read") and not l.startswith("s_waitcnt")])

# without the 4 v_mov (mass copies) -- results wrong, timing only
variant("without the 4 v_mov_b32", [l for l in body if not l.startswith("v_mov_b32")])

# uniform-mass form: drop the m*inv multiplies (every v_pk_mul with op_sel_hi:[0,1] = broadcast of the mass) and the v_mov
variant("uniform-mass form (no v_mov, no m*inv v_pk_mul)", [l for l in body if not l.startswith("v_mov_b32") and not (l.startswith("v_pk_mul_f32") and "op_sel_hi:[0,1]" in l)])

variant("uniform-mass form, VALU only",
        [l for l in body if not l.startswith("v_mov_b32") and not (l.startswith("v_pk_mul_f32") and "op_sel_hi:[0,1]" in l) and not l.startswith("ds_read") and not l.startswith("s_waitcnt")])

# rsq only / pk only decomposition
variant("only the 16 v_rsq_f32", [l for l in body if l.startswith("v_rsq")])
variant("only the v_pk_* (no rsq, no LDS)", [l for l in body if l.startswith("v_pk_")])
if len(loops) > 1:
    unit = loops[1][1]
    variant("unit loop, ds_read_b96 -> ds_read_b128", [re.sub(r"ds_read_b96 v\[(\d+):(\d+)\]", lambda m: f"ds_read_b128 v[{m.group(1)}:{int(m.group(2)) + 1}]", l) for l in unit])
    variant("unit loop without s_nop", [l for l in unit if not l.startswith("s_nop")])
    variant("unit loop, VALU only", [l for l in unit if not l.startswith("ds_read") and not l.startswith("s_waitcnt")])

if len(loops) > 1:
    unit = loops[1][1]
    get = ["v_lshrrev_b32 v1, 8, %4", "v_readfirstlane_b32 s41, v1"]  # s41 = wave >> 2 (0..3): position among the 4 mates of a SIMD
    def by_slot(p0, p1, p2, p3):
        return get + ["s_cmp_eq_u32 s41, 0", "s_cbranch_scc0 11f", f"s_setprio {p0}", "11:", "s_cmp_eq_u32 s41, 1", "s_cbranch_scc0 12f", f"s_setprio {p1}", "12:",
                      "s_cmp_eq_u32 s41, 2", "s_cbranch_scc0 13f", f"s_setprio {p2}", "13:", "s_cmp_eq_u32 s41, 3", "s_cbranch_scc0 14f", f"s_setprio {p3}", "14:"]
    variant("unit loop, static prio 0,1,2,3 by wave>>2", unit, prologue=by_slot(0, 1, 2, 3))
    variant("unit loop, static prio 3,2,1,0 by wave>>2", unit, prologue=by_slot(3, 2, 1, 0))
    variant("unit loop, all prio 3", unit, prologue=["s_setprio 3"])


# ---- body j from SGPRs: every lane of a wave reads the same body j, so its coordinates could be scalar operands (s_load through the
# scalar cache instead of an LDS ring); here only the operand form is timed -- no loads at all -- against "unit loop, VALU only"
if len(loops) > 1:
    unit = loops[1][1]
    lo = int(re.search(r"ds_read_b128 v\[(\d+):", next(l for l in unit if l.startswith("ds_read_b128"))).group(1))
    valu = [l for l in unit if not l.startswith("ds_read") and not l.startswith("s_waitcnt")]
    sets = ["s_mov_b32 s60, 0x3f800000", "s_mov_b32 s61, 0x40000000", "s_mov_b32 s62, 0x40400000", "s_mov_b32 s63, 0x3f800000", "s_mov_b32 s64, 0x3c23d70a", "s_mov_b32 s65, 0x3c23d70a"]
    def j_from_sgprs(l):
        if not l.startswith("v_pk_add_f32"):
            return l
        return re.sub(rf"^(v_pk_add_f32 v\[\d+:\d+\], )v\[{lo}:{lo + 1}\]", r"\1s[60:61]", re.sub(rf"^(v_pk_add_f32 v\[\d+:\d+\], )v\[{lo + 2}:{lo + 3}\]", r"\1s[62:63]", l))
    sg = [j_from_sgprs(l) for l in valu]
    variant("unit loop, VALU only, body j as SGPR operands", sg, prologue=sets)
    eps = next(re.search(r"(v\[\d+:\d+\])$", l).group(1) for l in valu if l.startswith("v_pk_fma_f32") and re.search(r"(v\[\d+:\d+\]), \1, v\[\d+:\d+\]$", l))
    variant("unit loop, VALU only, body j and softening as SGPR operands", [l.replace(", " + eps, ", s[64:65]") if l.startswith("v_pk_fma_f32") and l.endswith(eps) and re.search(r"(v\[\d+:\d+\]), \1, ", l) else l for l in sg], prologue=sets)
    variant("unit loop, VALU only (again, for drift)", valu)

# ---- pipe-overlap probes: do packed fp32, plain fp32 and transcendental ops share one issue pipe? (32 instr per iteration) ----
def acc(k):
    return 2 + 2 * (k % 24)          # accumulator pairs v[2..49]
PK  = lambda k: f"v_pk_fma_f32 v[{acc(k)}:{acc(k)+1}], v[100:101], v[102:103], v[{acc(k)}:{acc(k)+1}]"
FMA = lambda k: f"v_fma_f32 v{acc(k)}, v100, v103, v{acc(k)}"
FMAC = lambda k: f"v_fmac_f32 v{acc(k)}, v100, v103"
RSQ = lambda k: f"v_rsq_f32 v{acc(k)}, v{acc(k)}"
MUL = lambda k: f"v_mul_f32 v{acc(k)}, v100, v{acc(k)}"
variant("probe: 32 v_pk_fma_f32", [PK(k) for k in range(32)], pairs_per_iter=32)
variant("probe: 32 v_fma_f32", [FMA(k) for k in range(32)], pairs_per_iter=32)
variant("probe: 32 v_fmac_f32", [FMAC(k) for k in range(32)], pairs_per_iter=32)
variant("probe: 16 v_pk_fma + 16 v_fma_f32 interleaved", [PK(k) if k % 2 == 0 else FMA(k) for k in range(32)], pairs_per_iter=32)
variant("probe: 16 v_pk_fma + 16 v_fmac_f32 interleaved", [PK(k) if k % 2 == 0 else FMAC(k) for k in range(32)], pairs_per_iter=32)
variant("probe: 16 v_pk_fma then 16 v_fma_f32 (blocks)", [PK(k) for k in range(16)] + [FMA(k) for k in range(16, 32)], pairs_per_iter=32)
variant("probe: 8 v_rsq + 24 v_pk_fma interleaved", [RSQ(k) if k % 4 == 0 else PK(k) for k in range(32)], pairs_per_iter=32)
variant("probe: 8 v_rsq + 24 v_fma_f32 interleaved", [RSQ(k) if k % 4 == 0 else FMA(k) for k in range(32)], pairs_per_iter=32)
variant("probe: 32 v_rsq_f32", [RSQ(k) for k in range(32)], pairs_per_iter=32)
variant("probe: 32 v_mul_f32", [MUL(k) for k in range(32)], pairs_per_iter=32)
# fp64 issue costs (what the FAST fp64 loop is made of): v[100:103] hold finite doubles made of the float patterns, results go nowhere useful
D = lambda k: 2 + 2 * (k % 24)
variant("probe: 32 v_fma_f64", [f"v_fma_f64 v[{D(k)}:{D(k)+1}], v[100:101], v[102:103], v[{D(k)}:{D(k)+1}]" for k in range(32)], pairs_per_iter=32)
variant("probe: 32 v_mul_f64", [f"v_mul_f64 v[{D(k)}:{D(k)+1}], v[100:101], v[{D(k)}:{D(k)+1}]" for k in range(32)], pairs_per_iter=32)
variant("probe: 32 v_add_f64", [f"v_add_f64 v[{D(k)}:{D(k)+1}], v[100:101], v[{D(k)}:{D(k)+1}]" for k in range(32)], pairs_per_iter=32)
variant("probe: 32 v_rsq_f64", [f"v_rsq_f64 v[{D(k)}:{D(k)+1}], v[100:101]" for k in range(32)], pairs_per_iter=32)
variant("probe: 32 v_rcp_f64", [f"v_rcp_f64 v[{D(k)}:{D(k)+1}], v[100:101]" for k in range(32)], pairs_per_iter=32)
variant("probe: 32 v_cvt_f32_f64", [f"v_cvt_f32_f64 v{D(k)}, v[100:101]" for k in range(32)], pairs_per_iter=32)
variant("probe: 32 v_cvt_f64_f32", [f"v_cvt_f64_f32 v[{D(k)}:{D(k)+1}], v100" for k in range(32)], pairs_per_iter=32)
variant("probe: 8 v_rsq_f64 + 24 v_fma_f64 interleaved", [f"v_rsq_f64 v[{D(k)}:{D(k)+1}], v[100:101]" if k % 4 == 0 else f"v_fma_f64 v[{D(k)}:{D(k)+1}], v[100:101], v[102:103], v[{D(k)}:{D(k)+1}]" for k in range(32)], pairs_per_iter=32)
# packed ops with ONE scalar source, by operand position (s[60:65] are set by the prologue)
SETS = ["s_mov_b32 s60, 0x3f800000", "s_mov_b32 s61, 0x40000000", "s_mov_b32 s62, 0x40400000", "s_mov_b32 s63, 0x3f800000", "s_mov_b32 s64, 0x3c23d70a", "s_mov_b32 s65, 0x3c23d70a"]
variant("probe: 32 v_pk_add_f32 v,v", [f"v_pk_add_f32 v[{acc(k)}:{acc(k)+1}], v[100:101], v[{acc(k)}:{acc(k)+1}]" for k in range(32)], pairs_per_iter=32)
variant("probe: 32 v_pk_add_f32 s,v (src0 scalar)", [f"v_pk_add_f32 v[{acc(k)}:{acc(k)+1}], s[60:61], v[{acc(k)}:{acc(k)+1}]" for k in range(32)], pairs_per_iter=32, prologue=SETS)
variant("probe: 32 v_pk_mul_f32 s,v (src0 scalar)", [f"v_pk_mul_f32 v[{acc(k)}:{acc(k)+1}], s[60:61], v[{acc(k)}:{acc(k)+1}]" for k in range(32)], pairs_per_iter=32, prologue=SETS)
variant("probe: 32 v_pk_fma_f32 v,v,s (src2 scalar)", [f"v_pk_fma_f32 v[{acc(k)}:{acc(k)+1}], v[100:101], v[{acc(k)}:{acc(k)+1}], s[64:65]" for k in range(32)], pairs_per_iter=32, prologue=SETS)
variant("probe: 32 v_pk_fma_f32 s,v,v (src0 scalar)", [f"v_pk_fma_f32 v[{acc(k)}:{acc(k)+1}], s[64:65], v[100:101], v[{acc(k)}:{acc(k)+1}]" for k in range(32)], pairs_per_iter=32, prologue=SETS)
variant("probe: 32 v_pk_fma_f32 v,v,v three distinct pairs", [f"v_pk_fma_f32 v[{acc(k)}:{acc(k)+1}], v[100:101], v[102:103], v[{acc(k)}:{acc(k)+1}]" for k in range(32)], pairs_per_iter=32)
variant("probe: 32 v_pk_fma_f32 v,v,0.5 (inline constant src2)", [f"v_pk_fma_f32 v[{acc(k)}:{acc(k)+1}], v[100:101], v[{acc(k)}:{acc(k)+1}], 0.5 op_sel_hi:[1,1,0]" for k in range(32)], pairs_per_iter=32)
variant("probe: 16 s_mov_b32 + 32 v_pk_fma_f32 (scalar moves interleaved)", [x for k in range(32) for x in ([PK(k)] + ([f"s_mov_b32 s{66 + (k % 8)}, s60"] if k % 2 == 0 else []))], pairs_per_iter=32, prologue=SETS)


# ---- s_setprio semantics: the two older waves of every SIMD (wave < 8) at level a, the two younger ones at level b -------
if len(loops) > 1:
    unit = loops[1][1]
    for a in range(4):
        for b in range(4):
            pro = ["v_lshrrev_b32 v1, 9, %4", "v_readfirstlane_b32 s41, v1", "s_cmp_eq_u32 s41, 0", "s_cbranch_scc0 11f", f"s_setprio {a}", "s_branch 12f", "11:", f"s_setprio {b}", "12:"]
            variant(f"setprio: waves 0-7 level {a}, waves 8-15 level {b}", unit, prologue=pro)

print("// GENERATED by tools/loop_microbench_gen.py -- do not edit")
print("#include <hip/hip_runtime.h>\n#include <algorithm>\n#include <cstdio>\n#include <cstdlib>\n#include <vector>")
clob = ", ".join(f'"v{r}"' for r in range(1, 128))
for idx, (name, lines, _, prologue) in enumerate(variants):
    print(f"__global__ __launch_bounds__(1024) void k{idx}(unsigned long long* out) {{")
    print("    extern __shared__ float4 lds[];")
    print("    lds[threadIdx.x] = make_float4(threadIdx.x, 1.f, 2.f, 1.f);")
    print("    lds[threadIdx.x + 1024] = make_float4(threadIdx.x, 1.f, 2.f, 1.f);")
    print("    lds[threadIdx.x + 2048] = make_float4(threadIdx.x, 1.f, 2.f, 1.f);")
    print("    lds[threadIdx.x + 3072] = make_float4(threadIdx.x, 1.f, 2.f, 1.f);")
    print("    __syncthreads();")
    print("    unsigned long long t0, t1, r0, r1;")
    print("    unsigned tid = threadIdx.x;")
    print("    asm volatile(")
    for r in range(1, 128):
        print(f'        "v_cvt_f32_u32 v{r}, %4\\n\\tv_mul_f32 v{r}, 0x3a83126f, v{r}\\n\\tv_add_f32 v{r}, 1.0, v{r}\\n\\t"')
    for l in prologue:
        print(f'        "{l}\\n\\t"')
    print(f'        "s_movk_i32 s40, {OUTER}\\n\\t"')
    print('        "s_memtime %0\\n\\t"')
    print('        "s_memrealtime %2\\n\\t"')
    print('        "s_waitcnt lgkmcnt(0)\\n"')
    print('        "2:\\n\\t"')
    addr = next((re.search(r", (v\d+)", l).group(1) for l in lines if l.startswith("ds_read")), "v45")  # the loop's LDS address VGPR
    print(f'        "v_lshrrev_b32 {addr}, 6, %4\\n\\t"')          # wave id -> slice base (2 KiB per wave)
    print(f'        "v_lshlrev_b32 {addr}, 11, {addr}\\n\\t"')
    print('        "s_movk_i32 s26, 32\\n"')
    print('        "1:\\n\\t"')
    for l in lines:
        print(f'        "{l}\\n\\t"')
    print('        "s_sub_u32 s26, s26, 1\\n\\t"')
    print('        "s_cmp_lg_u32 s26, 0\\n\\t"')
    print('        "s_cbranch_scc1 1b\\n\\t"')
    print('        "s_sub_u32 s40, s40, 1\\n\\t"')
    print('        "s_cmp_lg_u32 s40, 0\\n\\t"')
    print('        "s_cbranch_scc1 2b\\n\\t"')
    print('        "s_waitcnt lgkmcnt(0)\\n\\t"')
    print('        "s_memtime %1\\n\\t"')
    print('        "s_memrealtime %3\\n\\t"')
    print('        "s_waitcnt lgkmcnt(0)\\n\\t"')
    print(f'        : "=&s"(t0), "=&s"(t1), "=&s"(r0), "=&s"(r1) : "v"(tid) : {clob}, "s26", "s40", "s41", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "scc", "vcc", "memory");')
    print("    if ((threadIdx.x & 63) == 0) {")
    print("        unsigned long long* o = out + (blockIdx.x * 16 + (threadIdx.x >> 6)) * 4;")
    print("        o[0] = t0, o[1] = t1, o[2] = r0, o[3] = r1;")
    print("    }")
    print("}")
print("struct T { const char* name; void (*k)(unsigned long long*); int pairs; };")
print("static T tests[] = {")
for idx, (name, _, pairs, _p) in enumerate(variants):
    print(f'    {{"{name}", k{idx}, {pairs}}},')
print("};")
print(f"constexpr int OUTER = {OUTER};")
print(r"""
static float run(void (*k)(unsigned long long*), int cus, int threads, unsigned long long* out, hipEvent_t e0, hipEvent_t e1, int warm) {
    for (int r = 0; r < warm; ++r) hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 64 * 1024, 0, out);  // reach the clock the chip holds under this load
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 64 * 1024, 0, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    unsigned long long* out;
    if (hipMalloc(&out, sizeof(unsigned long long) * cus * 16 * 4) != hipSuccess) return 1;
    std::vector<unsigned long long> h(cus * 16 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    int index = 0;
    for (auto& t : tests) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(t.k), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        // the second test (the production unit-mass loop) is also run with 1, 2 and 3 waves per SIMD
        for (int wps = (index == 1 ? 1 : 4); wps <= 4; ++wps) {
            const float ms = run(t.k, cus, 256 * wps, out, e0, e1, 3);
            (void)hipMemcpy(h.data(), out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            const int nw = 4 * wps;
            // per workgroup (= per CU): busy span in shader cycles and in 100 MHz ticks
            std::vector<double> span, clock;
            for (int b = 0; b < cus; ++b) {
                unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0;
                for (int w = 0; w < nw; ++w) {
                    const unsigned long long* o = &h[(b * 16 + w) * 4];
                    t0 = std::min(t0, o[0]), t1 = std::max(t1, o[1]), r0 = std::min(r0, o[2]), r1 = std::max(r1, o[3]);
                }
                span.push_back(static_cast<double>(t1 - t0));
                clock.push_back(static_cast<double>(t1 - t0) / static_cast<double>(r1 - r0) * 0.1);  // GHz
            }
            std::sort(span.begin(), span.end());
            std::sort(clock.begin(), clock.end());
            const double iters = 32.0 * OUTER;
            const double med   = span[span.size() / 2];
            // each of the CU's 4 SIMDs time-shares `wps` waves: SIMD cycles per packed pair = span / (iters * pairs * wps)
            printf("%-72s waves/SIMD %d  SIMD cycles/pair %6.2f (min %.2f max %.2f)  clock %.3f GHz -> %.2f ns/pair/SIMD  wall %.3f ms\n", t.name, wps, med / iters / t.pairs / wps,
                   span.front() / iters / t.pairs / wps, span.back() / iters / t.pairs / wps, clock[clock.size() / 2], med / iters / t.pairs / wps / clock[clock.size() / 2], ms);
            printf("      waves of workgroup 0, (t1 - min t0) / span:");
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int w = 0; w < nw; ++w) t0 = std::min(t0, h[w * 4]), t1 = std::max(t1, h[w * 4 + 1]);
            for (int w = 0; w < nw; ++w) printf(" %.2f", static_cast<double>(h[w * 4 + 1] - t0) / static_cast<double>(t1 - t0));
            printf("\n");
            fflush(stdout);
        }
        ++index;
    }
    return 0;
}
""")
