// Generalized-Hilbert ("Gilbert") curve index and block adjacency on the GPU.
// One thread per voxel walks the cuboid subdivision iteratively (the reference recursion,
// gilbert.py:68-272, is tail-recursive in every branch, so a loop carrying (origin, a, b, c, cur) suffices).
#include "common.h"

namespace jenga {
namespace {

struct V3 {
    long long x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ long long comp_sum(V3 a) { return a.x + a.y + a.z; }
__device__ __forceinline__ long long iabs(long long v) { return v < 0 ? -v : v; }
__device__ __forceinline__ long long isgn(long long v) { return (v > 0) - (v < 0); }
__device__ __forceinline__ V3 vsgn(V3 a) { return {isgn(a.x), isgn(a.y), isgn(a.z)}; }
// Python floor division by two (operands can be negative)
__device__ __forceinline__ long long half_floor(long long v) { return v >> 1; }
__device__ __forceinline__ V3 vhalf(V3 a) { return {half_floor(a.x), half_floor(a.y), half_floor(a.z)}; }

// is `p` inside the cuboid with origin o and (signed) extents a+b+c ?  (gilbert.py:44-65)
__device__ __forceinline__ bool inside(V3 p, V3 o, V3 a, V3 b, V3 c) {
    const V3 d = a + b + c;
    const bool okx = d.x < 0 ? (p.x <= o.x && p.x > o.x + d.x) : (p.x >= o.x && p.x < o.x + d.x);
    const bool oky = d.y < 0 ? (p.y <= o.y && p.y > o.y + d.y) : (p.y >= o.y && p.y < o.y + d.y);
    const bool okz = d.z < 0 ? (p.z <= o.z && p.z > o.z + d.z) : (p.z >= o.z && p.z < o.z + d.z);
    return okx && oky && okz;
}

__device__ long long gilbert_index(long long px, long long py, long long pz, long long W, long long Hh, long long Dd) {
    const V3 p = {px, py, pz};
    V3 o = {0, 0, 0}, a, b, c;
    if (W >= Hh && W >= Dd) {  // gilbert.py:19-38: the longest axis leads
        a = {W, 0, 0}; b = {0, Hh, 0}; c = {0, 0, Dd};
    } else if (Hh >= W && Hh >= Dd) {
        a = {0, Hh, 0}; b = {W, 0, 0}; c = {0, 0, Dd};
    } else {
        a = {0, 0, Dd}; b = {W, 0, 0}; c = {0, Hh, 0};
    }
    long long cur = 0;
    for (int guard = 0; guard < 256; ++guard) {
        const long long w = iabs(comp_sum(a)), h = iabs(comp_sum(b)), d = iabs(comp_sum(c));
        const V3 da = vsgn(a), db = vsgn(b), dc = vsgn(c);
        const V3 rel = p - o;
        if (h == 1 && d == 1) return cur + da.x * rel.x + da.y * rel.y + da.z * rel.z;
        if (w == 1 && d == 1) return cur + db.x * rel.x + db.y * rel.y + db.z * rel.z;
        if (w == 1 && h == 1) return cur + dc.x * rel.x + dc.y * rel.y + dc.z * rel.z;

        V3 a2 = vhalf(a), b2 = vhalf(b), c2 = vhalf(c);
        const long long w2 = iabs(comp_sum(a2)), h2 = iabs(comp_sum(b2)), d2 = iabs(comp_sum(c2));
        if ((w2 & 1) && w > 2) a2 = a2 + da;  // prefer even steps
        if ((h2 & 1) && h > 2) b2 = b2 + db;
        if ((d2 & 1) && d > 2) c2 = c2 + dc;

        V3 no, na, nb, nc;  // next sub-cuboid
        if (2 * w > 3 * h && 2 * w > 3 * d) {  // wide: split along a only
            if (inside(p, o, a2, b, c)) {
                no = o; na = a2; nb = b; nc = c;
            } else {
                cur += iabs(comp_sum(a2) * comp_sum(b) * comp_sum(c));
                no = o + a2; na = a - a2; nb = b; nc = c;
            }
        } else if (3 * h > 4 * d) {  // split a,b ; keep c whole
            if (inside(p, o, b2, c, a2)) {
                no = o; na = b2; nb = c; nc = a2;
            } else {
                cur += iabs(comp_sum(b2) * comp_sum(c) * comp_sum(a2));
                if (inside(p, o + b2, a, b - b2, c)) {
                    no = o + b2; na = a; nb = b - b2; nc = c;
                } else {
                    cur += iabs(comp_sum(a) * comp_sum(b - b2) * comp_sum(c));
                    no = o + (a - da) + (b2 - db); na = -b2; nb = c; nc = -(a - a2);
                }
            }
        } else if (3 * d > 4 * h) {  // split a,c ; keep b whole
            if (inside(p, o, c2, a2, b)) {
                no = o; na = c2; nb = a2; nc = b;
            } else {
                cur += iabs(comp_sum(c2) * comp_sum(a2) * comp_sum(b));
                if (inside(p, o + c2, a, b, c - c2)) {
                    no = o + c2; na = a; nb = b; nc = c - c2;
                } else {
                    cur += iabs(comp_sum(a) * comp_sum(b) * comp_sum(c - c2));
                    no = o + (a - da) + (c2 - dc); na = -c2; nb = -(a - a2); nc = b;
                }
            }
        } else {  // regular: five sub-cuboids
            if (inside(p, o, b2, c2, a2)) {
                no = o; na = b2; nb = c2; nc = a2;
            } else {
                cur += iabs(comp_sum(b2) * comp_sum(c2) * comp_sum(a2));
                const V3 o1 = o + b2;
                if (inside(p, o1, c, a2, b - b2)) {
                    no = o1; na = c; nb = a2; nc = b - b2;
                } else {
                    cur += iabs(comp_sum(c) * comp_sum(a2) * comp_sum(b - b2));
                    const V3 o2 = o + (b2 - db) + (c - dc);
                    if (inside(p, o2, a, -b2, -(c - c2))) {
                        no = o2; na = a; nb = -b2; nc = -(c - c2);
                    } else {
                        cur += iabs(comp_sum(a) * comp_sum(-b2) * comp_sum(-(c - c2)));
                        const V3 o3 = o + (a - da) + b2 + (c - dc);
                        if (inside(p, o3, -c, -(a - a2), b - b2)) {
                            no = o3; na = -c; nb = -(a - a2); nc = b - b2;
                        } else {
                            cur += iabs(comp_sum(-c) * comp_sum(-(a - a2)) * comp_sum(b - b2));
                            no = o + (a - da) + (b2 - db); na = -b2; nb = c2; nc = -(a - a2);
                        }
                    }
                }
            }
        }
        o = no; a = na; b = nb; c = nc;
    }
    return -1;  // unreachable for valid input
}

__global__ void gilbert_map_kernel(int t, int h, int w, int64_t* __restrict__ l2h, int64_t* __restrict__ h2l) {
    const long long n = (long long)t * h * w;
    for (long long lin = blockIdx.x * (long long)blockDim.x + threadIdx.x; lin < n;
         lin += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(lin % w), y = (int)((lin / w) % h), z = (int)(lin / ((long long)w * h));
        const long long g = gilbert_index(x, y, z, w, h, t);
        l2h[lin] = g;
        h2l[g] = lin;
    }
}

// Sliced variant (gilbert.py:332-440): per-frame 2-D curve.  The flip of frame z depends on where frame z-1
// ended, which depends on frame z-1's own flip -> a tiny serial scan over t done by every thread (t <= a few
// dozen).  The un-flipped 2-D curve always ends at (ex0, ey0) = position of index h*w-1.
__global__ void gilbert_sliced_kernel(int t, int h, int w, int64_t* __restrict__ l2h, int64_t* __restrict__ h2l) {
    const long long sp = (long long)h * w, n = sp * t;
    __shared__ int end_x0, end_y0;
    if (threadIdx.x == 0) {
        // find the voxel holding the last index of the un-flipped slice: it is one of the 4 corners or lies on
        // the boundary; scan the whole slice cooperatively below instead of guessing.
        end_x0 = -1;
        end_y0 = -1;
    }
    __syncthreads();
    for (long long i = threadIdx.x; i < sp; i += blockDim.x) {
        const int x = (int)(i % w), y = (int)(i / w);
        if (gilbert_index(x, y, 0, w, h, 1) == sp - 1) {
            end_x0 = x;
            end_y0 = y;
        }
    }
    __syncthreads();
    const int ex0 = end_x0, ey0 = end_y0;
    for (long long lin = blockIdx.x * (long long)blockDim.x + threadIdx.x; lin < n;
         lin += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(lin % w), y = (int)((lin / w) % h), z = (int)(lin / sp);
        // replay the flip chain up to frame z
        bool fx = false, fy = false;
        for (int zz = 1; zz <= z; ++zz) {
            // frame zz-1 used flips (fx,fy): its last voxel in LOCAL (unflipped-loop) coordinates is the (x,y)
            // whose actual position is (ex0,ey0): x = fx ? w-1-ex0 : ex0
            const int lx = fx ? w - 1 - ex0 : ex0, ly = fy ? h - 1 - ey0 : ey0;
            fx = (2 * lx >= w);
            fy = (2 * ly >= h);
        }
        const int ax = fx ? w - 1 - x : x, ay = fy ? h - 1 - y : y;
        const long long g = (long long)z * sp + gilbert_index(ax, ay, 0, w, h, 1);
        l2h[lin] = g;
        h2l[g] = lin;
    }
}

__global__ void neighbors_kernel(int t, int h, int w, int block, const int64_t* __restrict__ l2h,
                                 uint8_t* __restrict__ out, long long nb) {
    const long long n = (long long)t * h * w;
    for (long long lin = blockIdx.x * (long long)blockDim.x + threadIdx.x; lin < n;
         lin += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(lin % w), y = (int)((lin / w) % h), z = (int)(lin / ((long long)w * h));
        const long long cb = l2h[lin] / block;
        for (int dz = -1; dz <= 1; ++dz) {
            const int nz = z + dz;
            if (nz < 0 || nz >= t) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                const int ny = y + dy;
                if (ny < 0 || ny >= h) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    const int nx = x + dx;
                    if (nx < 0 || nx >= w) continue;
                    const long long ob = l2h[((long long)nz * h + ny) * w + nx] / block;
                    out[cb * nb + ob] = 1;  // benign race: every writer stores the same byte
                }
            }
        }
    }
}

}  // namespace
}  // namespace jenga

using namespace jenga;

extern "C" int jenga_gilbert_map(void* stream, int t, int h, int w, int sliced, int64_t* l2h, int64_t* h2l) {
    if (t <= 0 || h <= 0 || w <= 0 || !l2h || !h2l) {
        set_error("jenga_gilbert_map: bad arguments t=%d h=%d w=%d", t, h, w);
        return JENGA_EINVAL;
    }
    const long long n = (long long)t * h * w;
    const int threads = 256;
    const int blocks = (int)((n + threads - 1) / threads > 4096 ? 4096 : (n + threads - 1) / threads);
    if (sliced)
        hipLaunchKernelGGL(gilbert_sliced_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, t, h, w, l2h,
                           h2l);
    else
        hipLaunchKernelGGL(gilbert_map_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, t, h, w, l2h, h2l);
    if (hipGetLastError() != hipSuccess) {
        set_error("jenga_gilbert_map: launch failed");
        return JENGA_ELAUNCH;
    }
    return JENGA_OK;
}

extern "C" int jenga_gilbert_neighbors(void* stream, int t, int h, int w, int block, const int64_t* l2h,
                                       uint8_t* out) {
    if (t <= 0 || h <= 0 || w <= 0 || block <= 0 || !l2h || !out) {
        set_error("jenga_gilbert_neighbors: bad arguments");
        return JENGA_EINVAL;
    }
    const long long n = (long long)t * h * w, nb = (n + block - 1) / block;
    const int threads = 256;
    const int blocks = (int)((n + threads - 1) / threads > 4096 ? 4096 : (n + threads - 1) / threads);
    hipLaunchKernelGGL(neighbors_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, t, h, w, block, l2h, out,
                       nb);
    if (hipGetLastError() != hipSuccess) {
        set_error("jenga_gilbert_neighbors: launch failed");
        return JENGA_ELAUNCH;
    }
    return JENGA_OK;
}
