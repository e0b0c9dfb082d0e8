// image_io.h -- PNG / Radiance-HDR I/O of the host entry point (see image_io.cpp).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace imgio {

struct Image {
    int width = 0, height = 0;
    std::vector<uint32_t> rgba; // r | g << 8 | b << 16 | a << 24, row 0 first
};

void write_png_rgba8(const std::string& path, int w, int h, const uint32_t* rgba);
Image load_png_rgba8(const std::string& path);
Image load_hdr_as_ldr_rgba8(const std::string& path);
void flip_vertical(Image& img);

} // namespace imgio
