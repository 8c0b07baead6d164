// A client of the C ABI that knows nothing of Python or PyTorch: stereo matching of one frame
// (Hamming 1-NN + the rectified-stereo row filter) followed by the triangulation of the kept
// matches -- find_stereo_matches + get_3d_points_cloud of the reference (utils/utils.py:302-335)
// -- with plain hipMalloc / hipMemcpy around slam_pack_descriptors, slam_hamming_match and
// slam_triangulate_matched.
//
//   stereo_pair <input file> <output file>
//
// input  (little endian): int32 nL, nR, desc_bytes, 0; uint8 descL[nL][desc_bytes], descR[nR][desc_bytes];
//                         float kpL[nL][2], kpR[nR][2]; double P[12], Q[12]; double row_threshold
// output: int32 idx[nL][2], dist[nL][2]; uint8 keep[nL]; double X[nL], Y[nL], Z[nL]; uint8 valid[nL];
//         int32 stereo_map[nL]
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "slam_amd.h"

#define HIP_OK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));         \
            return 2;                                                                               \
        }                                                                                           \
    } while (0)
#define SLAM_OK_OR_DIE(x)                                                                           \
    do {                                                                                            \
        if ((x) != SLAM_OK) {                                                                       \
            std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, slam_last_error());             \
            return 3;                                                                               \
        }                                                                                           \
    } while (0)

template <class T>
static bool read_vec(std::FILE *f, std::vector<T> &v, size_t n)
{
    v.resize(n);
    return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n;
}

template <class T>
static int to_device(const std::vector<T> &h, T **d)
{
    HIP_OK(hipMalloc(reinterpret_cast<void **>(d), std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty())
        HIP_OK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

template <class T>
static int from_device(std::vector<T> &h, const T *d, size_t n)
{
    h.resize(n);
    if (n)
        HIP_OK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 3) {
        std::fprintf(stderr, "usage: %s <input> <output>\n", argv[0]);
        return 1;
    }
    std::FILE *in = std::fopen(argv[1], "rb");
    if (!in) {
        std::perror(argv[1]);
        return 1;
    }
    int32_t hdr[4];
    if (std::fread(hdr, sizeof(int32_t), 4, in) != 4)
        return 1;
    const int nL = hdr[0], nR = hdr[1], D = hdr[2];
    if (nL < 0 || nR < 0 || D < 1 || D > 64)
        return 1;
    std::vector<uint8_t> descL, descR;
    std::vector<float> kpL, kpR;
    std::vector<double> PQ, thr;
    if (!read_vec(in, descL, (size_t)nL * D) || !read_vec(in, descR, (size_t)nR * D) || !read_vec(in, kpL, (size_t)nL * 2) ||
        !read_vec(in, kpR, (size_t)nR * 2) || !read_vec(in, PQ, 24) || !read_vec(in, thr, 1)) {
        std::fprintf(stderr, "short input file\n");
        return 1;
    }
    std::fclose(in);

    // SoA keypoint columns, as the library reads them
    std::vector<float> xL(nL), yL(nL), xR(nR), yR(nR);
    for (int i = 0; i < nL; ++i) {
        xL[i] = kpL[2 * i];
        yL[i] = kpL[2 * i + 1];
    }
    for (int i = 0; i < nR; ++i) {
        xR[i] = kpR[2 * i];
        yR[i] = kpR[2 * i + 1];
    }
    const int stride = D <= 32 ? 32 : 64;
    uint8_t *d_rawL, *d_rawR, *d_L, *d_R, *d_keep, *d_valid;
    float *d_xL, *d_yL, *d_xR, *d_yR;
    int32_t *d_seg, *d_idx, *d_dist, *d_map;
    double *d_X, *d_Y, *d_Z;
    if (to_device(descL, &d_rawL) || to_device(descR, &d_rawR) || to_device(xL, &d_xL) || to_device(yL, &d_yL) ||
        to_device(xR, &d_xR) || to_device(yR, &d_yR))
        return 2;
    const std::vector<int32_t> seg = {0, nL, 0, nR, 0};   // one item: all left rows against all right rows
    if (to_device(seg, &d_seg))
        return 2;
    const size_t rows = std::max(nL, 1);
    HIP_OK(hipMalloc((void **)&d_L, std::max<size_t>((size_t)nL * stride, 64)));
    HIP_OK(hipMalloc((void **)&d_R, std::max<size_t>((size_t)nR * stride, 64)));
    HIP_OK(hipMalloc((void **)&d_idx, rows * 2 * sizeof(int32_t)));
    HIP_OK(hipMalloc((void **)&d_dist, rows * 2 * sizeof(int32_t)));
    HIP_OK(hipMalloc((void **)&d_keep, rows));
    HIP_OK(hipMalloc((void **)&d_valid, rows));
    HIP_OK(hipMalloc((void **)&d_map, rows * sizeof(int32_t)));
    HIP_OK(hipMalloc((void **)&d_X, rows * sizeof(double)));
    HIP_OK(hipMalloc((void **)&d_Y, rows * sizeof(double)));
    HIP_OK(hipMalloc((void **)&d_Z, rows * sizeof(double)));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    SLAM_OK_OR_DIE(slam_pack_descriptors(d_rawL, nL, D, D, d_L, stride, stream));
    SLAM_OK_OR_DIE(slam_pack_descriptors(d_rawR, nR, D, D, d_R, stride, stream));
    SLAM_OK_OR_DIE(slam_hamming_match(d_L, d_R, d_seg, 1, nL, nR, D, stride, 1, 2, thr[0], d_yL, d_yR, d_idx, d_dist,
                                      d_keep, stream));
    SLAM_OK_OR_DIE(slam_triangulate_matched(PQ.data(), PQ.data() + 12, d_xL, d_yL, d_xR, d_yR, d_seg, 1, nL, d_idx, d_keep,
                                            d_X, d_Y, d_Z, d_valid, d_map, stream));
    HIP_OK(hipStreamSynchronize(stream));

    std::vector<int32_t> idx, dist, map;
    std::vector<uint8_t> keep, valid;
    std::vector<double> X, Y, Z;
    if (from_device(idx, d_idx, (size_t)nL * 2) || from_device(dist, d_dist, (size_t)nL * 2) || from_device(keep, d_keep, nL) ||
        from_device(X, d_X, nL) || from_device(Y, d_Y, nL) || from_device(Z, d_Z, nL) || from_device(valid, d_valid, nL) ||
        from_device(map, d_map, nL))
        return 2;
    std::FILE *out = std::fopen(argv[2], "wb");
    if (!out) {
        std::perror(argv[2]);
        return 1;
    }
    std::fwrite(idx.data(), sizeof(int32_t), idx.size(), out);
    std::fwrite(dist.data(), sizeof(int32_t), dist.size(), out);
    std::fwrite(keep.data(), 1, keep.size(), out);
    std::fwrite(X.data(), sizeof(double), X.size(), out);
    std::fwrite(Y.data(), sizeof(double), Y.size(), out);
    std::fwrite(Z.data(), sizeof(double), Z.size(), out);
    std::fwrite(valid.data(), 1, valid.size(), out);
    std::fwrite(map.data(), sizeof(int32_t), map.size(), out);
    std::fclose(out);
    size_t kept = 0, good = 0;
    for (int i = 0; i < nL; ++i) {
        kept += keep[i];
        good += valid[i];
    }
    std::printf("left keypoints %d, right keypoints %d, matches kept %zu, points in front of the cameras %zu (library version %d)\n",
                nL, nR, kept, good, slam_version());
    for (void *p : {(void *)d_rawL, (void *)d_rawR, (void *)d_L, (void *)d_R, (void *)d_keep, (void *)d_valid, (void *)d_xL,
                    (void *)d_yL, (void *)d_xR, (void *)d_yR, (void *)d_seg, (void *)d_idx, (void *)d_dist, (void *)d_map,
                    (void *)d_X, (void *)d_Y, (void *)d_Z})
        HIP_OK(hipFree(p));
    HIP_OK(hipStreamDestroy(stream));
    return 0;
}
