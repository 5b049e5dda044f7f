// image_pipeline.hip -- the reference's image input pipeline on the device:
//   train: Resize(resize) -> RandomCrop(crop) -> RandomHorizontalFlip -> ToTensor -> Normalize
//   eval:  Resize((crop, crop)) -> ToTensor -> Normalize                 (ref: vilmedic/datasets/base/ImageDataset.py:96-108)
// on decoded uint8 HWC images already resident in HBM.  torchvision's Resize on PIL images is Pillow's 8-bit bilinear
// resampler (antialiasing triangle filter, 22-bit fixed-point coefficients, uint8 rounding after EACH of the two passes);
// that integer arithmetic is reproduced exactly: the per-axis coefficient tables are computed on the host with the same
// double-precision operations as Pillow's precompute_coeffs / normalize_coeffs_8bpc (only for the crop's rows and
// columns), and one fused kernel does horizontal pass -> uint8 -> vertical pass -> uint8 -> crop -> flip -> /255 ->
// (x - mean) / std for each output pixel.  HBM-bound byte work: every source byte is read from HBM once (taps of
// neighbouring outputs hit L1/L2), 12 B written per output pixel.
#include "common.h"
#include <cmath>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#define IP_PRECISION_BITS 22

struct ImgDesc { int64_t off; int32_t H, W, flip, top, left, pad; int64_t tx, ty; };   // tx / ty: int32 offsets of the axis tables

// one thread = one output pixel (3 channels).  tabs: per distinct (in, out) size pair one table [out][2 + KS] int32:
// (first source index, taps, coefficients); images of equal size share tables, a crop only offsets into them.
__global__ __launch_bounds__(256) void image_pipeline_kernel(const uint8_t* __restrict__ src, const ImgDesc* __restrict__ desc,
                                                             const int32_t* __restrict__ tabs, int KS, int crop,
                                                             float m0, float m1, float m2, float s0, float s1, float s2,
                                                             float* __restrict__ out) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= crop * crop) return;
    const int y = idx / crop, x = idx - y * crop;
    const ImgDesc d = desc[b];
    const int xs = d.flip ? crop - 1 - x : x;                       // column of the (unflipped) crop this output shows
    const int32_t* th = tabs + d.tx + (int64_t)(d.left + xs) * (2 + KS);
    const int32_t* tv = tabs + d.ty + (int64_t)(d.top + y) * (2 + KS);
    const int xmin = th[0], nx = th[1], ymin = tv[0], ny = tv[1];
    const uint8_t* img = src + d.off;
    const int half = 1 << (IP_PRECISION_BITS - 1);
    int a0 = half, a1 = half, a2 = half;
    for (int j = 0; j < ny; ++j) {
        const uint8_t* row = img + ((int64_t)(ymin + j) * d.W + xmin) * 3;
        int t0 = half, t1 = half, t2 = half;
        for (int i = 0; i < nx; ++i) {
            const int k = th[2 + i];
            t0 += row[3 * i] * k; t1 += row[3 * i + 1] * k; t2 += row[3 * i + 2] * k;
        }
        const int kv = tv[2 + j];                                   // horizontal pass result is rounded + clipped to uint8 first
        a0 += min(max(t0 >> IP_PRECISION_BITS, 0), 255) * kv;
        a1 += min(max(t1 >> IP_PRECISION_BITS, 0), 255) * kv;
        a2 += min(max(t2 >> IP_PRECISION_BITS, 0), 255) * kv;
    }
    const float f0 = (float)min(max(a0 >> IP_PRECISION_BITS, 0), 255) / 255.0f;
    const float f1 = (float)min(max(a1 >> IP_PRECISION_BITS, 0), 255) / 255.0f;
    const float f2 = (float)min(max(a2 >> IP_PRECISION_BITS, 0), 255) / 255.0f;
    const int64_t plane = (int64_t)crop * crop;
    float* o = out + (int64_t)b * 3 * plane + idx;
    o[0] = (f0 - m0) / s0;
    o[plane] = (f1 - m1) / s1;
    o[2 * plane] = (f2 - m2) / s2;
}

// ---------------------------------------------------------------- host: Pillow's coefficient tables (Resample.c, bilinear)
#pragma clang fp contract(off)
static double ip_bilinear(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}
struct AxisTable { int ksize; std::vector<int32_t> bounds, kk; };      // bounds [out][2], kk [out][ksize]
static AxisTable ip_precompute(int in_size, int out_size) {
    AxisTable t;
    const double scale = (double)in_size / out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    t.ksize = (int)ceil(support) * 2 + 1;
    t.bounds.assign((size_t)out_size * 2, 0);
    t.kk.assign((size_t)out_size * t.ksize, 0);
    const double ss = 1.0 / filterscale;
    std::vector<double> k(t.ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = ip_bilinear((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            t.kk[(size_t)xx * t.ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << IP_PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << IP_PRECISION_BITS));
        }
        t.bounds[(size_t)xx * 2] = xmin;
        t.bounds[(size_t)xx * 2 + 1] = xmax;
    }
    return t;
}

extern "C" size_t vm_image_pipeline_ws(int B, int max_out, int max_taps) {
    // worst case: every image has its own pair of tables, each max_out rows (max_out = largest resized side in the batch)
    return (size_t)B * (sizeof(ImgDesc) + (size_t)2 * max_out * (2 + max_taps) * sizeof(int32_t));
}

// pinned staging buffer for the descriptors + tables (reused across calls; an event guards the previous call's copy)
static void* g_ip_host = nullptr;
static size_t g_ip_host_bytes = 0;
static hipEvent_t g_ip_event = nullptr;

extern "C" int vm_image_pipeline_u8(const uint8_t* src, const int64_t* src_offset, const int32_t* src_hw, int B, int resize, int crop,
                                    const int32_t* crop_top_left, const uint8_t* flip, const float* mean, const float* stdv,
                                    float* out, int max_taps, void* ws, size_t ws_bytes, void* stream) {
    VM_REQUIRE(src && src_offset && src_hw && mean && stdv && out && ws, "vm_image_pipeline_u8: null pointer");
    VM_REQUIRE(B > 0 && crop > 0 && resize >= 0 && max_taps >= 3, "vm_image_pipeline_u8: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const int stride = 2 + max_taps;
    std::vector<ImgDesc> descs(B);
    std::vector<int32_t> tabs;                                   // distinct tables, back to back
    std::map<std::pair<int, int>, int64_t> where;               // (in, out) -> int32 offset in tabs
    int bad_taps = 0;
    auto table = [&](int in_size, int out_size) -> int64_t {
        auto key = std::make_pair(in_size, out_size);
        auto it = where.find(key);
        if (it != where.end()) return it->second;
        const AxisTable t = ip_precompute(in_size, out_size);
        if (t.ksize > max_taps) { bad_taps = t.ksize; return 0; }
        const int64_t off = (int64_t)tabs.size();
        tabs.resize(tabs.size() + (size_t)out_size * stride, 0);
        for (int xx = 0; xx < out_size; ++xx) {
            int32_t* dst = tabs.data() + off + (size_t)xx * stride;
            dst[0] = t.bounds[(size_t)xx * 2]; dst[1] = t.bounds[(size_t)xx * 2 + 1];
            for (int k = 0; k < t.ksize; ++k) dst[2 + k] = t.kk[(size_t)xx * t.ksize + k];
        }
        where[key] = off;
        return off;
    };
    double bytes = 0;
    for (int b = 0; b < B; ++b) {
        const int H = src_hw[2 * b], W = src_hw[2 * b + 1];
        VM_REQUIRE(H > 0 && W > 0, "vm_image_pipeline_u8: image %d has size %dx%d", b, H, W);
        int nh, nw, top = 0, left = 0;
        if (resize > 0) {            // torchvision Resize(int): shorter side -> resize, longer -> int(resize * long / short)
            if (H <= W) { nh = resize; nw = (int)((double)resize * W / H); } else { nh = (int)((double)resize * H / W); nw = resize; }
            if (crop_top_left) { top = crop_top_left[2 * b]; left = crop_top_left[2 * b + 1]; }
            VM_REQUIRE(top >= 0 && left >= 0 && top + crop <= nh && left + crop <= nw,
                       "vm_image_pipeline_u8: crop window (%d,%d)+%d outside the %dx%d resized image %d", top, left, crop, nh, nw, b);
        } else { nh = crop; nw = crop; }
        ImgDesc& d = descs[b];
        d.off = src_offset[b]; d.H = H; d.W = W; d.flip = flip ? flip[b] != 0 : 0; d.top = top; d.left = left; d.pad = 0;
        d.tx = table(W, nw);
        d.ty = table(H, nh);
        VM_REQUIRE(bad_taps == 0, "vm_image_pipeline_u8: image %d needs %d taps (max_taps=%d)", b, bad_taps, max_taps);
        bytes += (double)H * W * 3 + 12.0 * crop * crop;
    }
    const size_t desc_bytes = (size_t)B * sizeof(ImgDesc), need = desc_bytes + tabs.size() * sizeof(int32_t);
    VM_REQUIRE(ws_bytes >= need, "vm_image_pipeline_u8: workspace of %zu bytes needed (vm_image_pipeline_ws gives the bound)", need);
    if (g_ip_event) hipEventSynchronize(g_ip_event); else hipEventCreateWithFlags(&g_ip_event, hipEventDisableTiming);
    if (g_ip_host_bytes < need) {
        if (g_ip_host) hipHostFree(g_ip_host);
        VM_REQUIRE(hipHostMalloc(&g_ip_host, need, hipHostMallocDefault) == hipSuccess, "vm_image_pipeline_u8: pinned staging allocation failed");
        g_ip_host_bytes = need;
    }
    memcpy(g_ip_host, descs.data(), desc_bytes);
    memcpy((char*)g_ip_host + desc_bytes, tabs.data(), tabs.size() * sizeof(int32_t));
    hipMemcpyAsync(ws, g_ip_host, need, hipMemcpyHostToDevice, s);
    hipEventRecord(g_ip_event, s);
    VmProfScope prof(VM_FAM_ELT, bytes, s, "image_pipeline_B%d_crop%d", B, crop);
    const ImgDesc* dd = (const ImgDesc*)ws;
    const int32_t* dt = (const int32_t*)((char*)ws + desc_bytes);
    hipLaunchKernelGGL(image_pipeline_kernel, dim3((crop * crop + 255) / 256, B), dim3(256), 0, s, src, dd, dt, max_taps, crop,
                       mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], out);
    return vm_check_launch("vm_image_pipeline_u8");
}
