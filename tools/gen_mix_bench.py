# tools/gen_mix_bench.py OUT.hip -> hipcc --offload-arch=gfx950 -O3 OUT.hip -o tools/bin/mix_bench   (profiles/r06_b_issue_classes.md)
# synthetic instruction streams: issue behaviour of mixes of fast / slow, VOP2 / VOP3 at 1, 2, 4 waves per SIMD
import itertools, sys
NREG=48  # destination registers v40..v87 rotate; sources v88..v99 never written -> no RAW at all except through dst reuse far apart
def F3(i): d=40+(i%NREG); return f"v_bitop3_b32 v{d}, v{88+i%4}, v{92+i%4}, v{96+i%4} bitop3:0x96"
def F2(i): d=40+(i%NREG); return f"v_xor_b32 v{d}, v{88+i%4}, v{92+i%4}"
def S(i): d=40+2*(i%(NREG//2)); return f"v_lshl_add_u64 v[{d}:{d+1}], v[{88+2*(i%2)}:{89+2*(i%2)}], 1, v[{92+2*(i%2)}:{93+2*(i%2)}]"
def SB(i): d=40+(i%NREG); return f"v_bfe_u32 v{d}, v{88+i%4}, 3, 1"
def stream(pattern, n=640):
    out=[]; i=0
    for kind in itertools.islice(itertools.cycle(pattern), n):
        out.append({'F':F3,'f':F2,'S':S,'b':SB}[kind](i)); i+=1
    return out
variants=[("all fast VOP3 (bitop3)", "F"),
          ("all fast VOP2 (xor e32)", "f"),
          ("all slow (lshl_add_u64)", "S"),
          ("F F F F S interleaved", "FFFFS"),
          ("32 F then 8 S (clustered)", "F"*32+"S"*8),
          ("128 F then 32 S (clustered)", "F"*128+"S"*32),
          ("f f f f S interleaved (VOP2 fast)", "ffffS"),
          ("F f alternating", "Ff"),
          ("F F f f S mix like the pass", "FFffS"),
          ("F S alternating", "FS"),
          ("F F F F b interleaved (bfe)", "FFFFb")]
src=['#include <hip/hip_runtime.h>','#include <cstdio>','#include <cstdint>','#include <vector>','#include <algorithm>',
'struct Stamp { uint64_t cyc, real; };',
'template <int V> __global__ __launch_bounds__(256) void k(unsigned* out, Stamp* st, int iters) {',
' extern __shared__ uint4 pin[];',
' const uint64_t c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();',
' for (int i = 0; i < iters; ++i) {']
clob=", ".join(f'"v{r}"' for r in range(40,100))
for vi,(name,pat) in enumerate(variants):
    body="".join(f'"{l}\\n\\t"\n' for l in stream(pat))
    src.append(f' if (V == {vi}) asm volatile({body} ::: {clob});')
src+=[' }',' const uint64_t c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();',
' out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)c1;',
' if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c1 - c0, t1 - t0};','}',
'static unsigned* g_out; static Stamp* g_st;',
'template <int V> static void row(const char* name, int wps) {',
' const int iters = 400, blocks = 256 * wps; const size_t lds = (size_t)(160 * 1024 / wps) & ~(size_t)255;',
' hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);',
' hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);',
' hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(256), lds, 0, g_out, g_st, 4); hipDeviceSynchronize();',
' hipEventRecord(e0); hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(256), lds, 0, g_out, g_st, iters); hipEventRecord(e1); hipEventSynchronize(e1);',
' float ms; hipEventElapsedTime(&ms, e0, e1);',
' std::vector<Stamp> st((size_t)blocks * 4); hipMemcpy(st.data(), g_st, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost);',
' std::vector<double> cyc, clk; for (auto& s : st) { cyc.push_back((double)s.cyc); clk.push_back(s.real ? (double)s.cyc / (double)s.real * 0.1 : 0.0); }',
' std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());',
' const double instr = 640.0 * iters, ghz = clk[clk.size() / 2];',
' printf("%-40s w=%d  wave %6.2f cyc/instr   SIMD(wall) %6.2f cyc/instr   clock %.2f GHz\\n", name, wps, cyc[cyc.size() / 2] / instr, ms * 1e-3 * ghz * 1e9 / (instr * wps), ghz);',
'}',
'int main() {',' hipMalloc(&g_out, 256 * 8 * 256 * 4); hipMalloc(&g_st, 256 * 8 * 4 * sizeof(Stamp));',
' for (int w : {1, 2, 4}) {']
for vi,(name,pat) in enumerate(variants):
    src.append(f'  row<{vi}>("{name}", w);')
src+=[' }',' return 0;','}']
open(sys.argv[1] if len(sys.argv) > 1 else 'mix_bench.hip','w').write("\n".join(src))
