// TEST INFRASTRUCTURE (not product code).  A stand-in for the C++ translation unit Cython generates from the
// reference's lib/flow_c/gpu_flow.pyx:13-16,35-40: it includes the reference's OWN header, unmodified, from where it
// lies (-I/root/reference/lib/flow_c, see oracle/Makefile) and calls `_flow` through it — so it links whatever symbol
// that header declares (C++ linkage: _Z5_flowPfS_S_S_S_S_iiii).  Linked against mx_deepim_amd/libdeepim_hip.so, the
// way INTEGRATION.md option 1 links the reference's `gpu_flow` extension.  Built into oracle/_ref/ (git-ignored).
#include "gpu_flow.hpp"

extern "C" void flow_hpp_client_call(float* flow, float* valid, float* depth_src, float* depth_tgt, float* KT,
                                     float* Kinv, int batch_size, int height, int width, int device_id) {
  _flow(flow, valid, depth_src, depth_tgt, KT, Kinv, batch_size, height, width, device_id);
}
