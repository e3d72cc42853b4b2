#!/usr/bin/env python3
"""bank effect on v_pk_* issue cost, long runs (wall clock; relative numbers).  python3 bank2_gen.py > bank2.hip"""
ITERS = 400000
tests = []
def pair(r): return f"v[{r}:{r+1}]"
def add(name, body):
    assert len(body) == 32
    tests.append((name, body))
for dc in (0, 2):
    for ac in (0, 2):
        for bc in (0, 2):
            add(f"pk_fma acc(%4={dc}) += a(%4={ac})*b(%4={bc})", [f"v_pk_fma_f32 {pair(4*(k%16)+dc)}, {pair(64+4*(k%8)+ac)}, {pair(96+4*(k%8)+bc)}, {pair(4*(k%16)+dc)}" for k in range(32)])
for dc in (0, 2):
    for ac in (0, 2):
        add(f"pk_fma w(%4={dc}) += x(%4={ac})^2", [f"v_pk_fma_f32 {pair(4*(k%16)+dc)}, {pair(64+4*(k%8)+ac)}, {pair(64+4*(k%8)+ac)}, {pair(4*(k%16)+dc)}" for k in range(32)])
for dc in (0, 2):
    for ac in (0, 2):
        for bc in (0, 2):
            add(f"pk_add d(%4={dc}) = a(%4={ac}).lo - b(%4={bc})", [f"v_pk_add_f32 {pair(4*(k%16)+dc)}, {pair(64+4*(k%8)+ac)}, {pair(96+4*(k%8)+bc)} op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" for k in range(32)])
for dc in (0, 2):
    for ac in (0, 2):
        for bc in (0, 2):
            add(f"pk_mul d(%4={dc}) = a(%4={ac})*b(%4={bc})", [f"v_pk_mul_f32 {pair(4*(k%16)+dc)}, {pair(64+4*(k%8)+ac)}, {pair(96+4*(k%8)+bc)}" for k in range(32)])
# odd-aligned? (pairs must be even-aligned on gfx90a+) -- classes are 0 and 2 only.
add("v_rsq_f32 x32", [f"v_rsq_f32 v{k%32}, v{64+k%32}" for k in range(32)])
add("v_mov_b32_dpp wave_ror x32", [f"v_mov_b32_dpp v{k%32}, v{64+k%32} wave_ror:1 row_mask:0xf bank_mask:0xf" for k in range(32)])
add("v_fma_f32 x32 three banks", [f"v_fma_f32 v{4*(k%16)}, v{64+4*(k%8)+1}, v{96+4*(k%8)+2}, v{4*(k%16)}" for k in range(32)])
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <vector>")
print(f"constexpr int ITERS = {ITERS};")
for i, (name, body) in enumerate(tests):
    print(f"__global__ __launch_bounds__(256) void k{i}(float* out) {{")
    print("    asm volatile(")
    print(f'        "s_mov_b32 s20, {ITERS}\\n"')
    print('        "1:\\n"')
    for ins in body:
        print(f'        "{ins}\\n"')
    print('        "s_sub_u32 s20, s20, 1\\n"')
    print('        "s_cmp_lg_u32 s20, 0\\n"')
    print('        "s_cbranch_scc1 1b\\n"')
    clob = ", ".join(f'"v{r}"' for r in range(128))
    print(f'        ::: "s20", "scc", {clob});')
    print("    if (out == nullptr) out[threadIdx.x] = 0;")
    print("}")
print("struct T { const char* name; void (*k)(float*); };")
print("int main() {")
print("    T tests[] = {" + ", ".join(f'{{"{n}", k{i}}}' for i, (n, _) in enumerate(tests)) + "};")
print("""    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& t : tests) {
        hipLaunchKernelGGL(t.k, dim3(256 * 4), dim3(256), 0, 0, out);  // 4 waves per SIMD
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(t.k, dim3(256 * 4), dim3(256), 0, 0, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-48s %8.3f ms  %6.3f ns per wave-instruction per SIMD  (= %5.2f cycles at 2.27 GHz)\\n", t.name, ms, ms * 1e6 / (32.0 * ITERS * 4), ms * 1e6 / (32.0 * ITERS * 4) * 2.27);
        fflush(stdout);
    }
    return 0;
}""")
