// odtk_binding.cpp -- the reference's operator module (NVIDIA/retinanet-examples csrc/extensions.cpp:47-158, :184-201)
// re-homed on the MI355X C ABI (include/odtk_hip.h, libodtk_hip.so): a compiled torch extension exporting
//
//     decode(cls_head, box_head, anchors, scale, score_thresh, top_n, rotated=False) -> [scores, boxes, classes]
//     nms(scores, boxes, classes, nms_thresh, detections_per_im, rotated=False)     -> [scores, boxes, classes]
//     iou(boxes, anchors)                                                           -> [Tensor[num_anchors, num_boxes]]
//     Engine                                                                        -> placeholder (TensorRT dropped)
//
// with the reference's positional signatures, plus `detect` (all pyramid levels + NMS in one enqueue).  It is what
// INTEGRATION.md section 2 tells a maintainer of the reference to write: no kernel lives here -- every function
// checks its tensors like the reference does (CUDA + contiguous -> RuntimeError), allocates outputs and scratch with
// torch, and makes the two-phase C ABI call on torch's current HIP stream with the GIL released.  Built by
// __graft_entry__.build() as retinanet-examples_amd/odtk/_C_ext*.so (plain host C++, g++; links libodtk_hip.so);
// `ODTK_BINDING=ext` makes odtk/_C.py hand decode / nms / iou to it instead of ctypes.
#include <torch/extension.h>
// PyTorch-ROCm keeps the device type named "cuda": the guard and stream accessors that honour that are the
// *MasqueradingAsCUDA forms (what hipify turns c10::cuda::CUDAGuard / getCurrentCUDAStream of the reference's
// extensions.cpp:62,99 into); the plain c10::hip:: forms reject a "cuda" device.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/odtk_hip.h"

namespace {

void require_gpu_contiguous(const torch::Tensor &t, const char *name) {       // extensions.cpp:42-44 CHECK_INPUT
  TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
  TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32");
}

int checked(int rc, const char *what) {
  if (rc < 0) {
    std::string msg = std::string(what) + " failed: ";
    switch (rc) {
      case ODTK_ERR_INVALID: msg += "invalid argument"; break;
      case ODTK_ERR_WORKSPACE: msg += "Workspace is too small!"; break;        // the reference's message, utils.h:55-57
      case ODTK_ERR_UNSUPPORTED: msg += "unsupported dtype/layout"; break;
      default: msg += std::string("HIP error (") + odtk_last_hip_error() + ")";
    }
    throw std::runtime_error(msg);
  }
  return rc;
}

void *current_stream(const torch::Tensor &t) {
  return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

// cub-style two-phase call (decode.cu:53-72): `call(workspace, size)` with (nullptr, 0) returns the bytes needed
template <typename Call>
void with_scratch(const torch::Tensor &like, const char *what, Call &&call) {
  const int size = checked(call(nullptr, 0), what);
  auto scratch = torch::empty({size > 0 ? size : 1}, like.options().dtype(torch::kUInt8));
  checked(call(scratch.data_ptr(), static_cast<size_t>(size)), what);
}

std::vector<torch::Tensor> decode(torch::Tensor cls_head, torch::Tensor box_head, std::vector<float> &anchors, int scale,
                                  float score_thresh, int top_n, bool rotated) {
  require_gpu_contiguous(cls_head, "cls_head");
  require_gpu_contiguous(box_head, "box_head");
  TORCH_CHECK(cls_head.dim() == 4 && box_head.dim() == 4 && !anchors.empty() && anchors.size() % 4 == 0, "decode: bad shapes");
  const int nb = rotated ? 6 : 4;
  const int64_t batch = cls_head.size(0), height = cls_head.size(2), width = cls_head.size(3);
  const size_t num_anchors = anchors.size() / 4, num_classes = cls_head.size(1) / num_anchors;
  TORCH_CHECK(box_head.size(0) == batch && box_head.size(1) == static_cast<int64_t>(num_anchors) * nb &&
              box_head.size(2) == height && box_head.size(3) == width, "box_head does not match cls_head");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(cls_head.device());
  auto opt = cls_head.options();
  auto scores = torch::empty({batch, top_n}, opt), boxes = torch::empty({batch, top_n, nb}, opt);
  auto classes = torch::empty({batch, top_n}, opt);
  const void *in[2] = {cls_head.data_ptr(), box_head.data_ptr()};
  void *out[3] = {scores.data_ptr(), boxes.data_ptr(), classes.data_ptr()};
  auto fn = rotated ? odtk_decode_rotate : odtk_decode;
  void *stream = current_stream(cls_head);
  {
    pybind11::gil_scoped_release nogil;
    with_scratch(cls_head, "decode", [&](void *ws, size_t size) {
      return fn(static_cast<int>(batch), ws ? in : nullptr, ws ? out : nullptr, height, width, scale, num_anchors, num_classes,
                anchors.data(), anchors.size(), score_thresh, top_n, ws, size, ws ? stream : nullptr);
    });
  }
  return {scores, boxes, classes};
}

std::vector<torch::Tensor> nms(torch::Tensor scores, torch::Tensor boxes, torch::Tensor classes, float nms_thresh,
                               int detections_per_im, bool rotated) {
  require_gpu_contiguous(scores, "scores");
  require_gpu_contiguous(boxes, "boxes");
  require_gpu_contiguous(classes, "classes");
  const int nb = rotated ? 6 : 4;
  TORCH_CHECK(scores.dim() == 2 && boxes.dim() == 3 && boxes.size(2) == nb && boxes.size(0) == scores.size(0) &&
              boxes.size(1) == scores.size(1) && classes.sizes() == scores.sizes(), "nms: inconsistent shapes");
  const int64_t batch = scores.size(0), count = scores.size(1);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(scores.device());
  auto opt = scores.options();
  auto out_scores = torch::empty({batch, detections_per_im}, opt), out_boxes = torch::empty({batch, detections_per_im, nb}, opt);
  auto out_classes = torch::empty({batch, detections_per_im}, opt);
  const void *in[3] = {scores.data_ptr(), boxes.data_ptr(), classes.data_ptr()};
  void *out[3] = {out_scores.data_ptr(), out_boxes.data_ptr(), out_classes.data_ptr()};
  auto fn = rotated ? odtk_nms_rotate : odtk_nms;
  void *stream = current_stream(scores);
  {
    pybind11::gil_scoped_release nogil;
    with_scratch(scores, "nms", [&](void *ws, size_t size) {
      return fn(static_cast<int>(batch), ws ? in : nullptr, ws ? out : nullptr, static_cast<size_t>(count), detections_per_im,
                nms_thresh, ws, size, ws ? stream : nullptr);
    });
  }
  return {out_scores, out_boxes, out_classes};
}

std::vector<torch::Tensor> iou(torch::Tensor boxes, torch::Tensor anchors) {
  require_gpu_contiguous(boxes, "boxes");
  require_gpu_contiguous(anchors, "anchors");
  const int num_boxes = static_cast<int>(boxes.numel() / 8), num_anchors = static_cast<int>(anchors.numel() / 8);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(boxes.device());
  auto out = torch::empty({num_anchors, num_boxes}, boxes.options());          // layout of extensions.cpp:64-66
  const void *in[2] = {boxes.data_ptr(), anchors.data_ptr()};
  void *outs[1] = {out.data_ptr()};
  void *stream = current_stream(boxes);
  {
    pybind11::gil_scoped_release nogil;
    checked(odtk_iou(in, outs, num_boxes, num_anchors, stream), "iou");
  }
  return {out};
}

// All pyramid levels + NMS in one enqueue (odtk_detect), for head tensors as the convolutions wrote them.
std::vector<torch::Tensor> detect(std::vector<torch::Tensor> cls_heads, std::vector<torch::Tensor> box_heads,
                                  std::vector<std::vector<float>> anchors, std::vector<int> strides, float score_thresh,
                                  int top_n, float nms_thresh, int detections_per_im, bool rotated, bool logits) {
  const size_t n = cls_heads.size();
  TORCH_CHECK(n >= 1 && n <= ODTK_MAX_LEVELS && box_heads.size() == n && anchors.size() == n && strides.size() == n,
              "detect: need 1..", ODTK_MAX_LEVELS, " levels with matching lists");
  const int nb = rotated ? 6 : 4;
  const auto dtype = cls_heads[0].scalar_type();
  const int dt = dtype == torch::kFloat32 ? ODTK_F32 : dtype == torch::kBFloat16 ? ODTK_BF16 : dtype == torch::kFloat16 ? ODTK_F16 : -1;
  TORCH_CHECK(dt >= 0, "detect: heads must be float32 / bfloat16 / float16");
  const int64_t batch = cls_heads[0].size(0);
  const int num_anchors = static_cast<int>(anchors[0].size() / 4);
  std::vector<odtk_level_t> levels(n);
  for (size_t i = 0; i < n; ++i) {
    const auto &c = cls_heads[i], &b = box_heads[i];
    TORCH_CHECK(c.is_cuda() && b.is_cuda() && c.dim() == 4 && b.dim() == 4 && c.scalar_type() == dtype && b.scalar_type() == dtype,
                "detect: level ", i, ": CUDA 4-d tensors of one dtype expected");
    // (a one-channel head or a 1 x 1 level is contiguous in BOTH readings and follows its partner: odtk/_C.py:_pair_layout)
    const bool c_nchw = c.is_contiguous(), c_nhwc = c.is_contiguous(at::MemoryFormat::ChannelsLast);
    const bool b_nchw = b.is_contiguous(), b_nhwc = b.is_contiguous(at::MemoryFormat::ChannelsLast);
    TORCH_CHECK(c_nchw || c_nhwc, "cls_head[", i, "] must be contiguous (NCHW or channels_last)");
    TORCH_CHECK((c_nchw && b_nchw) || (c_nhwc && b_nhwc), "box_head[", i, "] must share cls_head's memory format");
    const bool nchw = c_nchw && b_nchw;
    TORCH_CHECK(static_cast<int>(anchors[i].size()) == 4 * num_anchors && c.size(0) == batch && b.size(1) == num_anchors * nb, "detect: inconsistent level ", i);
    levels[i] = odtk_level_t{c.data_ptr(), b.data_ptr(), static_cast<int32_t>(c.size(2)), static_cast<int32_t>(c.size(3)), strides[i],
                             nchw ? 0 : 1, anchors[i].data(), nullptr, nullptr};
  }
  const int num_classes = static_cast<int>(cls_heads[0].size(1)) / num_anchors;
  c10::hip::HIPGuardMasqueradingAsCUDA guard(cls_heads[0].device());
  auto opt = cls_heads[0].options().dtype(torch::kFloat32);
  auto scores = torch::empty({batch, detections_per_im}, opt), boxes = torch::empty({batch, detections_per_im, nb}, opt);
  auto classes = torch::empty({batch, detections_per_im}, opt);
  void *out[3] = {scores.data_ptr(), boxes.data_ptr(), classes.data_ptr()};
  const uint32_t flags = (rotated ? ODTK_FLAG_ROTATED : 0u) | (logits ? ODTK_FLAG_LOGITS : 0u);
  void *stream = current_stream(cls_heads[0]);
  {
    pybind11::gil_scoped_release nogil;
    with_scratch(cls_heads[0], "detect", [&](void *ws, size_t size) {
      return odtk_detect(static_cast<int>(batch), static_cast<int>(n), levels.data(), num_anchors, num_classes, dt, flags, score_thresh,
                         top_n, nms_thresh, detections_per_im, ws ? out : nullptr, ws, size, ws ? stream : nullptr);
    });
  }
  return {scores, boxes, classes};
}

struct Engine {};   // csrc/engine.h: the TensorRT engine class -- not available on MI355X (BASELINE.json north_star)

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  namespace py = pybind11;
  m.doc() = "odtk._C on MI355X: decode / nms / iou of NVIDIA/retinanet-examples over libodtk_hip.so";
  // ABI guard: this module was compiled against ONE revision of include/odtk_hip.h; the library it found at load time must
  // have been compiled against the same struct layouts (a stale module would pass level arrays of the wrong stride)
  if (odtk_abi_struct_size(0) != static_cast<int>(sizeof(odtk_level_t)) ||
      odtk_abi_struct_size(1) != static_cast<int>(sizeof(odtk_snap_level_t)) ||
      odtk_abi_struct_size(2) != static_cast<int>(sizeof(odtk_snap_rot_level_t)))
    throw py::import_error("odtk._C_ext was compiled against another revision of include/odtk_hip.h than libodtk_hip.so "
                           "(struct sizes differ): rebuild it -- make -C retinanet-examples_amd/csrc");
  py::class_<Engine>(m, "Engine")
      .def(py::init([](py::args, py::kwargs) -> Engine * {
        throw std::runtime_error("odtk._C.Engine: the TensorRT engine path is not available on MI355X");
      }))
      .def_static("load", [](const std::string &) -> Engine * {
        throw std::runtime_error("odtk._C.Engine.load: the TensorRT engine path is not available on MI355X");
      });
  m.def("decode", &decode, py::arg("cls_head"), py::arg("box_head"), py::arg("anchors"), py::arg("scale"), py::arg("score_thresh"),
        py::arg("top_n"), py::arg("rotated") = false);
  m.def("nms", &nms, py::arg("scores"), py::arg("boxes"), py::arg("classes"), py::arg("nms_thresh"), py::arg("detections_per_im"),
        py::arg("rotated") = false);
  m.def("iou", &iou, py::arg("boxes"), py::arg("anchors"));
  m.def("detect", &detect, py::arg("cls_heads"), py::arg("box_heads"), py::arg("anchors"), py::arg("strides"), py::arg("score_thresh"),
        py::arg("top_n"), py::arg("nms_thresh"), py::arg("detections_per_im"), py::arg("rotated") = false, py::arg("logits") = false);
  m.def("version", [] { return std::string(odtk_version()); });
  // sizes of the ABI structs THIS module was compiled against: a module older than include/odtk_hip.h passes arrays of the
  // wrong stride (tests/test_compiled_binding.py compares them with the ctypes mirror on every CPU run)
  m.def("abi_sizes", [] {
    return py::dict(py::arg("level") = sizeof(odtk_level_t), py::arg("snap_level") = sizeof(odtk_snap_level_t),
                    py::arg("snap_rot_level") = sizeof(odtk_snap_rot_level_t));
  });
}
