// B2, the binding the reference itself uses: lib/flow_c/gpu_flow.hpp:1-3 declares `_flow` WITHOUT extern "C" and
// gpu_flow.pyx:13-16 is compiled as C++ (setup_linux.py:116-125, language="c++"), so the Cython extension links the
// Itanium-mangled symbol _Z5_flowPfS_S_S_S_S_iiii.  This translation unit exports exactly that symbol next to the
// C-linkage `_flow` of flow.hip; both forward to one body.  It must not see include/deepim_hip.h (a C-linkage and a
// C++-linkage function of the same name and signature cannot be declared in one translation unit).
extern "C" void deepim_flow_host(float* flow, float* valid, float* depth_src, float* depth_tgt, float* KT, float* Kinv,
                                 int batch_size, int height, int width, int device_id);

__attribute__((visibility("default"))) void _flow(float* flow, float* valid, float* depth_src, float* depth_tgt, float* KT,
                                                  float* Kinv, int batch_size, int height, int width, int device_id) {
  deepim_flow_host(flow, valid, depth_src, depth_tgt, KT, Kinv, batch_size, height, width, device_id);
}
