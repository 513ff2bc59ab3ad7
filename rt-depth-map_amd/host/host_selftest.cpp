// host_selftest.cpp -- exercises HIPMatcherCore / HIPMorphCore through the C ABI from C++.
//   host_selftest                 : no-GPU contract check (construction must fail loudly)
//   host_selftest <in.bin> <out.bin> W H D w : GPU run; in.bin = left||right (u8), out.bin = disp (s16)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hip_matcher_core.h"

int main(int argc, char** argv)
{
    rtdm::Rect none;
    if (argc < 7) {
        int ndev = 0;
        const int rc = rtdm_device_count(&ndev);
        rtdm::HIPMatcherCore m(none, none, 31, 13, 0, 10, 64, 64, 10, 100, 32, 1, 320, 240);
        std::printf("devices=%d device_count_rc=%d ctor_status=%d (%s)\n", ndev, rc, m.status(), m.statusText());
        std::vector<uint8_t> img(320 * 240, 7);
        std::vector<int16_t> out(320 * 240);
        const int c = m.compute(img.data(), 320, img.data(), 320, 240, 320, out.data(), 640);
        std::printf("compute_status=%d\n", c);
        // rectifier: identity maps over an 8x6 frame, crop = everything; gray of (r, g, b) = (x, x, x) is x
        const int RW = 8, RH = 6;
        std::vector<int16_t> map1((size_t)RW * RH * 2);
        std::vector<uint16_t> map2((size_t)RW * RH, 0);
        std::vector<uint8_t> rgb((size_t)RW * RH * 3), gl((size_t)RW * RH), gr((size_t)RW * RH);
        for (int y = 0; y < RH; ++y)
            for (int x = 0; x < RW; ++x) {
                map1[((size_t)y * RW + x) * 2] = (int16_t)x; map1[((size_t)y * RW + x) * 2 + 1] = (int16_t)y;
                for (int k = 0; k < 3; ++k) rgb[((size_t)y * RW + x) * 3 + k] = (uint8_t)(x * 9 + y);
            }
        rtdm::Rect full; full.width = RW; full.height = RH;
        rtdm::HIPRectifierCore rect(map1.data(), map2.data(), map1.data(), map2.data(), RH, RW, full);
        const int rs = rect.rectifyGray(rgb.data(), RW * 3, rgb.data(), RW * 3, gl.data(), RW, gr.data(), RW);
        std::printf("rectifier_status=%d rectify_status=%d\n", rect.status(), rs);
        if (ndev == 0)
            return (m.status() == RTDM_ERR_NO_DEVICE && c == RTDM_ERR_NO_DEVICE && rs == RTDM_ERR_NO_DEVICE) ? 0 : 1;
        bool same = rs == RTDM_OK;
        for (int y = 0; y < RH && same; ++y)
            for (int x = 0; x < RW; ++x) same = same && gl[(size_t)y * RW + x] == (uint8_t)(x * 9 + y) && gr[(size_t)y * RW + x] == gl[(size_t)y * RW + x];
        return (m.status() == RTDM_OK && c == RTDM_OK && same) ? 0 : 1;
    }
    const int W = std::atoi(argv[3]), H = std::atoi(argv[4]), D = std::atoi(argv[5]), w = std::atoi(argv[6]);
    std::vector<uint8_t> in((size_t)2 * W * H);
    std::vector<int16_t> out((size_t)W * H);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(in.data(), 1, in.size(), f) != in.size()) return 2;
    std::fclose(f);
    rtdm::HIPMatcherCore m(none, none, 31, w, 0, 10, D, D, 10, 100, 32, 1, W, H);
    if (m.status() != RTDM_OK) return 3;
    if (m.compute(in.data(), W, in.data() + (size_t)W * H, W, H, W, out.data(), (size_t)W * 2) != RTDM_OK) return 4;
    f = std::fopen(argv[2], "wb");
    if (!f || std::fwrite(out.data(), 2, out.size(), f) != out.size()) return 5;
    std::fclose(f);
    return 0;
}
