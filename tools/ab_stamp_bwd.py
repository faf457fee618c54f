"""Scratch build helper (not product code): mke_attr_cnn.hip with s_memtime stamps in the BACKWARD convolution kernel
(k_attr_conv<3, 32, true, false, 2>: two triples per wavefront) -> tools/ab/libstamp.so; read with tools/attr_stamps.py bwd."""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "multike_amd", "csrc")
s = open(os.path.join(C, "mke_attr_cnn.hip")).read()
def rep(old, new, count=1):
    global s
    assert old in s, old[:70]
    s = s.replace(old, new, count)
rep("struct ConvParams {", "__device__ unsigned long long g_stamps[2048 * 16];\n#define STAMP(i) do { if (BWD && (threadIdx.x & 63) == 0) g_stamps[(blockIdx.x * NW + (threadIdx.x >> 6)) % 2048 * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)\nstruct ConvParams {")
rep("  __shared__ float s_flat[DENSE ? NSLOT : 1][DENSE ? FS : 1];", "  STAMP(0);\n  __shared__ float s_flat[DENSE ? NSLOT : 1][DENSE ? FS : 1];")
rep("  float(*xs)[DPX] = s_x[slot];", "  STAMP(1);\n  float(*xs)[DPX] = s_x[slot];")
rep("    float raw[2][WPL];\n", "    if (ra == 0x7fffffff) STAMP(15);\n    STAMP(2);\n    float raw[2][WPL];\n")
rep("    // ---- batch-norm affine, stage x with zero pads", "    if (raw[0][0] == 1.2345e30f) STAMP(15);\n    STAMP(3);\n    // ---- batch-norm affine, stage x with zero pads")
rep("    // ---- conv1 ---------", "    STAMP(4);\n    // ---- conv1 ---------")
rep("    // ---- conv2 + width normalisation ---", "    STAMP(5);\n    // ---- conv2 + width normalisation ---")
rep("      // ---- width-normalisation backward, tanh', parameter gradients of conv2", "      if (nrm[0][0] == 1.2345e30f) STAMP(15);\n      STAMP(6);\n      // ---- width-normalisation backward, tanh', parameter gradients of conv2")
rep("      // ---- conv2 transposed -> dc1, tanh', parameter gradients of conv1", "      STAMP(7);\n      // ---- conv2 transposed -> dc1, tanh', parameter gradients of conv1")
rep("      // ---- conv1 transposed -> dx, batch-norm affine backward, attribute-row gradient", "      STAMP(8);\n      // ---- conv1 transposed -> dx, batch-norm affine backward, attribute-row gradient")
rep("  __syncthreads();  // the strips are reused by the block-level reduction below\n", "  STAMP(9);\n  __syncthreads();  // the strips are reused by the block-level reduction below\n  STAMP(10);\n")
rep("    float* dst = p.ws ? p.ws + (size_t)(blockIdx.x % CNN_WS_COPIES) * CNN_WS_STRIDE(d) : p.gparams;", "    STAMP(11);\n    float* dst = p.ws ? p.ws + (size_t)(blockIdx.x % CNN_WS_COPIES) * CNN_WS_STRIDE(d) : p.gparams;")
s = s.rstrip()
# final stamp at the very end of the kernel body: before the closing brace of the BWD block
rep("      atomic_add_f32(dst + 2 * d + threadIdx.x, v);\n    }\n", "      atomic_add_f32(dst + 2 * d + threadIdx.x, v);\n    }\n    STAMP(12);\n")
rep('extern "C" int mke_attr_conv_fwd(', 'extern "C" int mke_debug_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mke::g_stamps), sizeof(unsigned long long) * 1024 * 16); }\nextern "C" int mke_attr_conv_fwd(')
os.makedirs("/tmp/stampbuild", exist_ok=True)
open("/tmp/stampbuild/mke_attr_cnn_stamp.hip", "w").write(s + "\n")
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-math-errno".split()
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-I", C, "-I", os.path.join(ROOT, "include"), "-c", "/tmp/stampbuild/mke_attr_cnn_stamp.hip", "-o", "/tmp/stampbuild/attr_stamp.o"])
objs = [os.path.join(C, f) for f in os.listdir(C) if f.endswith(".o") and f != "mke_attr_cnn.o"]
os.makedirs(os.path.join(ROOT, "tools", "ab"), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["/tmp/stampbuild/attr_stamp.o", "-o", os.path.join(ROOT, "tools", "ab", "libstamp.so")])
print("built tools/ab/libstamp.so (backward stamps)")
