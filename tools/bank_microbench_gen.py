#!/usr/bin/env python3
"""tools/bank_microbench_gen.py -- generates tools/bank_microbench.hip: gfx950 VALU issue cost as a function of
WHICH physical VGPRs an instruction reads (explicit register numbers in inline asm), to find out whether the
4.2 -> 5.45 cycle spread of v_pk_fma_f32 (profiles/round1_valu_microbench.txt) follows register banks.

Each kernel is one asm block: a loop of 32 independent instructions over accumulators v[0..63] with read-only
sources in v[64..127]; 256-thread blocks, W blocks per CU (W waves per SIMD).  Output: cycles per wave-instruction
per SIMD at the nominal 2.4 GHz (relative numbers are what matter).

    python3 tools/bank_microbench_gen.py > tools/bank_microbench.hip
    hipcc -O3 --offload-arch=gfx950 tools/bank_microbench.hip -o tools/bank_microbench
"""
ITERS = 2048
tests = []  # (name, [instr...])


def pair(r):
    return f"v[{r}:{r + 1}]"


def add(name, body):
    assert len(body) == 32, (name, len(body))
    tests.append((name, body))


# ---- Q1: v_pk_fma_f32 acc += a*b, acc pair class dc, a class ac, b class bc (class = reg % 4 in {0,2}) -------------
for dc in (0, 2):
    for ac in (0, 2):
        for bc in (0, 2):
            body = []
            for k in range(32):
                d = 4 * (k % 16) + dc
                a = 64 + 4 * (k % 8) + ac
                b = 96 + 4 * (k % 8) + bc
                body.append(f"v_pk_fma_f32 {pair(d)}, {pair(a)}, {pair(b)}, {pair(d)}")
            add(f"pk_fma acc(d%4={dc}) += a(%4={ac})*b(%4={bc})", body)

# same but a, b fixed registers for all 32 instructions (operand reuse)
for ac, bc in ((0, 0), (0, 2)):
    body = [f"v_pk_fma_f32 {pair(4 * (k % 16))}, {pair(64 + ac)}, {pair(96 + bc)}, {pair(4 * (k % 16))}" for k in range(32)]
    add(f"pk_fma acc += A*B fixed A(%4={ac}) B(%4={bc})", body)

# ---- Q1b: non-accumulating form d = a*b + c, all distinct -----------------------------------------------------------
for cc in (0, 2):
    for ac in (0, 2):
        for bc in (0, 2):
            body = []
            for k in range(32):
                d = 4 * (k % 16)
                a = 64 + 4 * (k % 8) + ac
                b = 96 + 4 * (k % 8) + bc
                c = 64 + 32 - 4 - 4 * (k % 8) + cc if False else 64 + 4 * ((k + 3) % 8) + cc
                body.append(f"v_pk_fma_f32 {pair(d)}, {pair(a)}, {pair(b)}, {pair(c)}")
            add(f"pk_fma d(%4=0) = a(%4={ac})*b(%4={bc}) + c(%4={cc})", body)

# ---- Q2: squares d2 = x*x + d2 (two distinct) -----------------------------------------------------------------------
for dc in (0, 2):
    for ac in (0, 2):
        body = [f"v_pk_fma_f32 {pair(4 * (k % 16) + dc)}, {pair(64 + 4 * (k % 8) + ac)}, {pair(64 + 4 * (k % 8) + ac)}, {pair(4 * (k % 16) + dc)}" for k in range(32)]
        add(f"pk_fma d2(%4={dc}) += x(%4={ac})^2", body)

# ---- Q3: pk_mul / pk_add two sources ---------------------------------------------------------------------------------
for op in ("v_pk_mul_f32", "v_pk_add_f32"):
    for dc in (0, 2):
        for ac in (0, 2):
            for bc in (0, 2):
                body = [f"{op} {pair(4 * (k % 16) + dc)}, {pair(64 + 4 * (k % 8) + ac)}, {pair(96 + 4 * (k % 8) + bc)}" for k in range(32)]
                add(f"{op} d(%4={dc}) = a(%4={ac}) . b(%4={bc})", body)

# the kernel's dx = bj.x - px : broadcast low half of a, negate b
for ac in (0, 2):
    for bc in (0, 2):
        body = [f"v_pk_add_f32 {pair(4 * (k % 16))}, {pair(64 + 4 * (k % 8) + ac)}, {pair(96 + 4 * (k % 8) + bc)} op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" for k in range(32)]
        add(f"pk_add bcast d = a.lo(%4={ac}) - b(%4={bc})", body)

# ---- Q4: unpacked v_fma_f32 d = a*b + d with banks ------------------------------------------------------------------
for (ab, bb, name) in ((0, 0, "a,b same bank as d"), (1, 2, "a,b,d three banks"), (1, 1, "a=b bank, d other"), (0, 1, "a = d bank, b other")):
    body = []
    for k in range(32):
        d = 4 * (k % 16)
        a = 64 + 4 * (k % 8) + ab
        b = 96 + 4 * (k % 8) + bb
        body.append(f"v_fma_f32 v{d}, v{a}, v{b}, v{d}")
    add(f"v_fma_f32 acc += a*b ({name})", body)
for (ab, bb, name) in ((0, 0, "same bank"), (1, 2, "three banks")):
    body = [f"v_fmac_f32 v{4 * (k % 16)}, v{64 + 4 * (k % 8) + ab}, v{96 + 4 * (k % 8) + bb}" for k in range(32)]
    add(f"v_fmac_f32 vop2 ({name})", body)
    body = [f"v_mul_f32 v{4 * (k % 16)}, v{64 + 4 * (k % 8) + ab}, v{96 + 4 * (k % 8) + bb}" for k in range(32)]
    add(f"v_mul_f32 vop2 ({name})", body)

# ---- Q5: the real inner-loop mix, per packed pair: 3 pk_add + 3 pk_fma(sq) + 2 rsq + 3 pk_mul + 3 pk_fma(acc) ------
# two register assignments: "naive" (everything class 0/2 alternating as allocated sequentially) and "split" (accumulators
# class 0, temporaries class 2, j operands class 0)
def mix(acc_c, tmp_c, j_c, pos_c):
    body = []
    for p in range(2):  # two pairs -> 2 x 14 = 28 instr, pad with 4 more adds to reach 32
        ax, ay, az = (4 * (3 * p + c) + acc_c for c in range(3))
        t = [24 + 4 * (5 * p + c) + tmp_c for c in range(5)]  # dx dy dz d2 s
        jx, jz = 64 + 8 * p + j_c, 64 + 8 * p + 4 + j_c        # [x y] [z m] of body j
        px, py, pz = (96 + 4 * (3 * p + c) + pos_c for c in range(3))
        e2 = 124 + 0
        body += [
            f"v_pk_add_f32 {pair(t[0])}, {pair(jx)}, {pair(px)} op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]",
            f"v_pk_add_f32 {pair(t[1])}, {pair(jx)}, {pair(py)} op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
            f"v_pk_add_f32 {pair(t[2])}, {pair(jz)}, {pair(pz)} op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]",
            f"v_pk_fma_f32 {pair(t[3])}, {pair(t[0])}, {pair(t[0])}, {pair(e2)}",
            f"v_pk_fma_f32 {pair(t[3])}, {pair(t[1])}, {pair(t[1])}, {pair(t[3])}",
            f"v_pk_fma_f32 {pair(t[3])}, {pair(t[2])}, {pair(t[2])}, {pair(t[3])}",
            f"v_rsq_f32 v{t[3]}, v{t[3]}",
            f"v_rsq_f32 v{t[3] + 1}, v{t[3] + 1}",
            f"v_pk_mul_f32 {pair(t[4])}, {pair(t[3])}, {pair(t[3])}",
            f"v_pk_mul_f32 {pair(t[3])}, {pair(jz)}, {pair(t[3])} op_sel:[1,0]",
            f"v_pk_mul_f32 {pair(t[4])}, {pair(t[3])}, {pair(t[4])}",
            f"v_pk_fma_f32 {pair(ax)}, {pair(t[0])}, {pair(t[4])}, {pair(ax)}",
            f"v_pk_fma_f32 {pair(ay)}, {pair(t[1])}, {pair(t[4])}, {pair(ay)}",
            f"v_pk_fma_f32 {pair(az)}, {pair(t[2])}, {pair(t[4])}, {pair(az)}",
        ]
    body += [f"s_nop 0"] * 4
    return body


for acc_c, tmp_c, j_c, pos_c in ((0, 0, 0, 0), (0, 2, 0, 2), (0, 2, 0, 0), (0, 2, 2, 0), (2, 0, 0, 2), (0, 0, 2, 2)):
    add(f"mix 2 pairs (dependent chains!) acc%4={acc_c} tmp%4={tmp_c} j%4={j_c} pos%4={pos_c}", mix(acc_c, tmp_c, j_c, pos_c))

print("// GENERATED by tools/bank_microbench_gen.py -- do not edit")
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>")
print(f"constexpr int ITERS = {ITERS};")
clob = ", ".join(f'"v{r}"' for r in range(128))
for idx, (name, body) in enumerate(tests):
    print(f"__global__ __launch_bounds__(256) void k{idx}(float* out) {{")
    print("    asm volatile(")
    for r in range(1, 128):
        print(f'        "v_cvt_f32_u32 v{r}, v0\\n\\tv_mul_f32 v{r}, 0x3a83126f, v{r}\\n\\tv_add_f32 v{r}, 1.0, v{r}\\n\\t"')  # 1 + tid*1e-3
    print('        "v_mov_b32 v0, 1.0\\n\\t"')
    print(f'        "s_movk_i32 s20, {ITERS}\\n"')
    print('        "1:\\n\\t"')
    for ins in body:
        print(f'        "{ins}\\n\\t"')
    print('        "s_sub_u32 s20, s20, 1\\n\\t"')
    print('        "s_cmp_lg_u32 s20, 0\\n\\t"')
    print('        "s_cbranch_scc1 1b\\n\\t"')
    print(f'        ::: {clob}, "s20", "scc", "memory");')
    print("    if (out == nullptr) out[0] = 1.0f;")
    print("}")
print("struct T { const char* name; void (*k)(float*); };")
print("static T tests[] = {")
for idx, (name, _) in enumerate(tests):
    print(f'    {{"{name}", k{idx}}},')
print("};")
print(r"""
int main(int argc, char** argv) {
    int only = argc > 1 ? atoi(argv[1]) : -1;
    float* out;
    if (hipMalloc(&out, 4) != hipSuccess) return 1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int idx = 0;
    for (auto& t : tests) {
        if (only >= 0 && idx++ != only) continue;
        printf("%-72s", t.name);
        for (int w : {1, 2, 4}) {
            const int blocks = cus * w;
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 3;
            printf("  w=%d %6.3f", w, ms * 1e-3 * 2.4e9 / (32.0 * ITERS * w));
        }
        printf("   cyc/instr/SIMD@2.4GHz\n");
        fflush(stdout);
    }
    return 0;
}
""")
