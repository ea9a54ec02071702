// TORCH_LIBRARY(open3d, ...) shim: the four torch ops the reference model calls (models/common_torch.py:127,133;
// models/v0/net_definitions_torch.py:30,107) registered from C++ with the Open3D v0.14.1 schemas, so that a libtorch /
// TorchScript caller -- asr::ReconstructSurface runs model.pt through module.run_method (cpp/lib/asr.cpp:315-326) --
// reaches the HIP kernels without a Python interpreter: load libasr_open3d_ops.so (torch.ops.load_library in Python,
// dlopen / link in C++) next to libasr_hip.so.  Every op unpacks pointers and sizes and calls the C ABI
// (include/asr_hip.h) on torch's current stream; one library context per device, held under a mutex per op.  Same argument checks as the Python
// registration (open3d/ml/torch/ops.py), which must not be imported into the same process (one definition per schema).
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include <map>
#include <mutex>
#include <tuple>

#include "../../../include/asr_hip.h"

namespace {

// ONE library context per device, shared by all calling threads and held under a mutex for the duration of an op: a
// context -- its stream, arenas and error text -- is not thread safe, and libtorch callers may run ops of one GPU from
// several threads (jit fork, multi-threaded C++ programs).  (Per-thread contexts multiplied the arenas, the auxiliary
// stream and the packed-weight cache by the number of threads.)  The contexts are never destroyed: at process exit the HIP
// runtime may already be gone when static destructors run.
struct DeviceContext {
    std::mutex mu;
    asr_hip_context* ctx = nullptr;
    hipStream_t last_stream = nullptr;  // stream of the previous op on this context
    hipEvent_t switch_ev = nullptr;
};
struct Locked {
    std::unique_lock<std::mutex> lock;
    asr_hip_context* c;
};
Locked context_for(const at::Tensor& t) {
    static std::mutex table_mu;
    static std::map<int, DeviceContext*>* table = new std::map<int, DeviceContext*>();  // leaked on purpose
    TORCH_CHECK(t.is_cuda(), "open3d ops (asr_hip): tensors must live on the GPU");
    const int dev = t.get_device();
    DeviceContext* dc;
    {
        std::lock_guard<std::mutex> g(table_mu);
        DeviceContext*& slot = (*table)[dev];
        if (!slot) slot = new DeviceContext();
        dc = slot;
    }
    std::unique_lock<std::mutex> lock(dc->mu);
    auto stream = at::hip::getCurrentHIPStream(dev).stream();
    c10::DeviceGuard guard(t.device());  // context, event and the record / wait below all belong to the tensor's device
    const bool first_use = dc->ctx == nullptr;
    if (first_use)
        TORCH_CHECK(asr_hip_context_create(&dc->ctx, stream) == ASR_HIP_OK, "asr_hip_context_create failed");
    // Threads on different streams share the context's scratch arena and flag pool: when the stream changes, the new stream
    // waits for everything the previous ops enqueued on the old one (ops are enqueued under the mutex, so an event recorded
    // now is behind all of them).  Nothing to wait for on the first use.
    if (!first_use && dc->last_stream != stream) {
        if (!dc->switch_ev)
            TORCH_CHECK(hipEventCreateWithFlags(&dc->switch_ev, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
        TORCH_CHECK(hipEventRecord(dc->switch_ev, dc->last_stream) == hipSuccess, "hipEventRecord failed");
        TORCH_CHECK(hipStreamWaitEvent(stream, dc->switch_ev, 0) == hipSuccess, "hipStreamWaitEvent failed");
    }
    dc->last_stream = stream;
    asr_hip_context_set_stream(dc->ctx, stream);
    return Locked{std::move(lock), dc->ctx};
}

void check(asr_hip_context* c, int rc, const char* what) {
    TORCH_CHECK(rc == ASR_HIP_OK, what, ": ", asr_hip_last_error(c));
}

at::Tensor dev_as(const at::Tensor& t, at::ScalarType dt, const at::Device& dev) {
    return t.to(dev, dt).contiguous();
}

at::Tensor sparse_conv(const at::Tensor& filters, const at::Tensor& inp_features, const at::Tensor& inp_importance,
                       const at::Tensor& neighbors_index, const at::Tensor& neighbors_kernel_index,
                       const at::Tensor& neighbors_importance, const at::Tensor& neighbors_row_splits, bool normalize,
                       int64_t /*max_temp_mem_MB*/) {
    TORCH_CHECK(inp_importance.numel() == 0, "sparse_conv: per-point inp_importance is not supported");
    TORCH_CHECK(filters.dim() == 3, "sparse_conv: filters must be [K, cin, cout]");
    const auto dev = filters.device();
    const at::Tensor w = dev_as(filters, at::kFloat, dev), f = dev_as(inp_features, at::kFloat, dev);
    const at::Tensor idx = dev_as(neighbors_index, at::kInt, dev), kidx = dev_as(neighbors_kernel_index, at::kByte, dev);
    const at::Tensor rs = dev_as(neighbors_row_splits, at::kLong, dev);
    at::Tensor nimp;
    if (neighbors_importance.numel()) nimp = dev_as(neighbors_importance, at::kFloat, dev);
    TORCH_CHECK(f.dim() == 2 && f.size(1) == w.size(1), "sparse_conv: feature width does not match the filter");
    const int64_t v = rs.size(0) - 1;
    at::Tensor out = at::empty({v, w.size(2)}, f.options());
    asr_sparse_conv_args a = {};  // zero: optional fields inactive
    a.filters = w.data_ptr<float>();
    a.inp_features = f.data_ptr<float>();
    a.inp_ld = f.stride(0);
    a.neighbors_importance = nimp.defined() ? nimp.data_ptr<float>() : nullptr;
    a.neighbors_index = idx.data_ptr<int32_t>();
    a.neighbors_kernel_index = kidx.data_ptr<uint8_t>();
    a.neighbors_row_splits = rs.data_ptr<int64_t>();
    a.num_out = v;
    a.num_inp = f.size(0);
    a.kernel_size = (int)w.size(0);
    a.cin = (int)w.size(1);
    a.cout = (int)w.size(2);
    a.normalize = normalize ? 1 : 0;
    a.out = out.data_ptr<float>();
    a.out_ld = out.stride(0);
    Locked held = context_for(w);  // released when the op returns
    asr_hip_context* c = held.c;
    check(c, asr_hip_sparse_conv_f32(c, &a), "sparse_conv");
    return out;
}

at::Tensor continuous_conv(const at::Tensor& filters, const at::Tensor& out_positions, const at::Tensor& extents,
                           const at::Tensor& offset, const at::Tensor& inp_positions, const at::Tensor& inp_features,
                           const at::Tensor& inp_importance, const at::Tensor& neighbors_index,
                           const at::Tensor& neighbors_importance, const at::Tensor& neighbors_row_splits,
                           bool align_corners, std::string coordinate_mapping, bool normalize, std::string interpolation,
                           int64_t /*max_temp_mem_MB*/) {
    TORCH_CHECK(align_corners && coordinate_mapping == "ball_to_cube_radial" && interpolation == "linear",
                "continuous_conv: only align_corners=True, ball_to_cube_radial, linear");
    TORCH_CHECK(inp_importance.numel() == 0, "continuous_conv: per-point inp_importance is not supported");
    TORCH_CHECK(offset.numel() == 0 || !offset.ne(0).any().item<bool>(), "continuous_conv: non-zero offset is not supported");
    TORCH_CHECK(filters.dim() == 5 && filters.size(0) == 4 && filters.size(1) == 4 && filters.size(2) == 4,
                "continuous_conv: only kernel_size [4,4,4] is implemented");
    const auto dev = filters.device();
    const at::Tensor w = dev_as(filters, at::kFloat, dev), op = dev_as(out_positions, at::kFloat, dev);
    const int64_t v = op.size(0);
    at::Tensor ext = dev_as(extents, at::kFloat, dev).reshape({-1});
    if (ext.size(0) == 1 && v != 1) ext = ext.expand({v}).contiguous();
    TORCH_CHECK(ext.size(0) == v, "continuous_conv: extents must be a scalar or have one entry per output");
    const at::Tensor ip = dev_as(inp_positions, at::kFloat, dev), f = dev_as(inp_features, at::kFloat, dev);
    TORCH_CHECK(f.dim() == 2 && f.size(1) == w.size(3), "continuous_conv: feature width does not match the filter");
    const at::Tensor idx = dev_as(neighbors_index, at::kInt, dev), rs = dev_as(neighbors_row_splits, at::kLong, dev);
    at::Tensor nimp;
    if (neighbors_importance.numel()) nimp = dev_as(neighbors_importance, at::kFloat, dev);
    at::Tensor out = at::empty({v, w.size(4)}, f.options());
    Locked held = context_for(w);  // released when the op returns
    asr_hip_context* c = held.c;
    check(c,
          asr_hip_continuous_conv_f32(c, w.data_ptr<float>(), op.data_ptr<float>(), ext.data_ptr<float>(),
                                      ip.data_ptr<float>(), f.data_ptr<float>(), idx.data_ptr<int32_t>(),
                                      nimp.defined() ? nimp.data_ptr<float>() : nullptr, rs.data_ptr<int64_t>(), v,
                                      (int)w.size(3), (int)w.size(4), normalize ? 1 : 0, nullptr, 0, out.data_ptr<float>()),
          "continuous_conv");
    return out;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> invert_neighbors_list(int64_t num_points, const at::Tensor& inp_index,
                                                                     const at::Tensor& inp_row_splits,
                                                                     const at::Tensor& inp_attributes) {
    const auto dev = inp_index.device();
    const at::Tensor idx = dev_as(inp_index, at::kInt, dev), rs = dev_as(inp_row_splits, at::kLong, dev);
    const bool has_attr = inp_attributes.numel() > 0;
    at::Tensor attr;
    if (has_attr) attr = dev_as(inp_attributes, at::kByte, dev);
    at::Tensor out_idx = at::empty({idx.size(0)}, idx.options());
    at::Tensor out_rs = at::empty({num_points + 1}, rs.options());
    at::Tensor out_attr = has_attr ? at::empty({idx.size(0)}, attr.options()) : at::empty({0}, inp_attributes.options());
    Locked held = context_for(idx);  // released when the op returns
    asr_hip_context* c = held.c;
    check(c,
          asr_hip_invert_neighbors_list(c, num_points, idx.data_ptr<int32_t>(), rs.data_ptr<int64_t>(), rs.size(0) - 1,
                                        has_attr ? attr.data_ptr<uint8_t>() : nullptr, out_idx.data_ptr<int32_t>(),
                                        out_rs.data_ptr<int64_t>(), has_attr ? out_attr.data_ptr<uint8_t>() : nullptr),
          "invert_neighbors_list");
    if (has_attr && inp_attributes.scalar_type() != at::kByte) out_attr = out_attr.to(inp_attributes.scalar_type());
    return std::make_tuple(out_idx.to(inp_index.scalar_type()), out_rs, out_attr);
}

at::Tensor reduce_subarrays_sum(const at::Tensor& values, const at::Tensor& row_splits) {
    const auto dev = values.device();
    const at::Tensor val = dev_as(values, at::kFloat, dev), rs = dev_as(row_splits, at::kLong, dev);
    const int64_t rows = rs.size(0) - 1;
    at::Tensor out = at::empty({rows}, val.options());
    Locked held = context_for(val);  // released when the op returns
    asr_hip_context* c = held.c;
    check(c, asr_hip_reduce_subarrays_sum(c, val.data_ptr<float>(), nullptr, rs.data_ptr<int64_t>(), rows, out.data_ptr<float>()),
          "reduce_subarrays_sum");
    return out;
}

}  // namespace

TORCH_LIBRARY(open3d, m) {
    m.def("invert_neighbors_list(int num_points, Tensor inp_neighbors_index, Tensor inp_neighbors_row_splits, "
          "Tensor inp_neighbors_attributes) -> (Tensor neighbors_index, Tensor neighbors_row_splits, "
          "Tensor neighbors_attributes)");
    m.def("reduce_subarrays_sum(Tensor values, Tensor row_splits) -> Tensor");
    m.def("sparse_conv(Tensor filters, Tensor inp_features, Tensor inp_importance, Tensor neighbors_index, "
          "Tensor neighbors_kernel_index, Tensor neighbors_importance, Tensor neighbors_row_splits, "
          "bool normalize=False, int max_temp_mem_MB=64) -> Tensor");
    m.def("continuous_conv(Tensor filters, Tensor out_positions, Tensor extents, Tensor offset, Tensor inp_positions, "
          "Tensor inp_features, Tensor inp_importance, Tensor neighbors_index, Tensor neighbors_importance, "
          "Tensor neighbors_row_splits, bool align_corners=False, str coordinate_mapping=\"ball_to_cube_radial\", "
          "bool normalize=False, str interpolation=\"linear\", int max_temp_mem_MB=64) -> Tensor");
}

// no CPU kernels: like the Python registration, the product has no CPU path
TORCH_LIBRARY_IMPL(open3d, CUDA, m) {
    m.impl("invert_neighbors_list", invert_neighbors_list);
    m.impl("reduce_subarrays_sum", reduce_subarrays_sum);
    m.impl("sparse_conv", sparse_conv);
    m.impl("continuous_conv", continuous_conv);
}
