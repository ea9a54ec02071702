// A libtorch program with no Python in the process: loads the op library, loads a TorchScript archive, runs its
// forward() (no arguments: the test module carries its inputs as buffers) on the GPU and prints the result as raw f32
// bytes to the file given as third argument.  usage: shim_demo <libasr_open3d_ops.so> <module.pt> <out.bin>
// (cpp/lib/asr.cpp:138-139,315-326 is the reference's version of this: torch::jit::load + run_method)
#include <dlfcn.h>
#include <torch/script.h>

#include <cstdio>

int main(int argc, char** argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: shim_demo <op library> <module.pt> <out.bin>\n");
        return 2;
    }
    if (!dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL)) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 1;
    }
    torch::jit::script::Module m = torch::jit::load(argv[2], torch::kCUDA);
    at::Tensor out = m.forward({}).toTensor().to(torch::kCPU, torch::kFloat).contiguous();
    FILE* f = fopen(argv[3], "wb");
    if (!f) return 1;
    fwrite(out.data_ptr<float>(), sizeof(float), (size_t)out.numel(), f);
    fclose(f);
    printf("rows %lld cols %lld\n", (long long)out.size(0), (long long)(out.dim() > 1 ? out.size(1) : 1));
    return 0;
}
