// image_io.cpp -- PNG / Radiance-HDR I/O for the host entry point (replaces stb_image / stb_image_write as used by
// path_tracer/src/utils/image_buffer.cpp:25-58 and application.cpp:225-246).  Own minimal codecs: the vendored stb headers are
// third-party code, not reference code, and are not copied.
#include "image_io.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace {

uint32_t crc_table[256];
bool crc_ready = false;
uint32_t crc32(uint32_t crc, const uint8_t* p, size_t n)
{
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[i] = c;
        }
        crc_ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data)
{
    put32(out, (uint32_t)data.size());
    std::vector<uint8_t> td(type, type + 4);
    td.insert(td.end(), data.begin(), data.end());
    out.insert(out.end(), td.begin(), td.end());
    put32(out, crc32(0, td.data(), td.size()));
}

// ---- inflate (RFC 1951), table-free canonical Huffman decoding ----
struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    int bits(int need)
    {
        uint32_t val = bitbuf;
        while (bitcnt < need) {
            if (pos >= n) throw std::runtime_error("png: truncated deflate stream");
            val |= (uint32_t)p[pos++] << bitcnt;
            bitcnt += 8;
        }
        bitbuf = need == 32 ? 0 : val >> need;
        bitcnt -= need;
        return (int)(val & ((need == 32) ? 0xFFFFFFFFu : ((1u << need) - 1)));
    }
};
struct Huffman {
    short count[16];
    short symbol[288];
    void build(const short* length, int n)
    {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[length[i]]++;
        short offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; ++len) offs[len + 1] = offs[len] + count[len];
        for (int i = 0; i < n; ++i)
            if (length[i]) symbol[offs[length[i]]++] = (short)i;
        count[0] = 0;
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; ++len) {
            code |= br.bits(1);
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        throw std::runtime_error("png: bad huffman code");
    }
};
void inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out)
{
    static const short lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const short lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const short dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const short dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    BitReader br{src, n};
    int last;
    do {
        last = br.bits(1);
        int type = br.bits(2);
        if (type == 0) {
            br.bitbuf = 0;
            br.bitcnt = 0;
            if (br.pos + 4 > n) throw std::runtime_error("png: truncated stored block");
            unsigned len = src[br.pos] | (src[br.pos + 1] << 8);
            br.pos += 4;
            if (br.pos + len > n) throw std::runtime_error("png: truncated stored block");
            out.insert(out.end(), src + br.pos, src + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lencode, distcode;
            short lengths[320];
            if (type == 1) {
                int s = 0;
                for (; s < 144; ++s) lengths[s] = 8;
                for (; s < 256; ++s) lengths[s] = 9;
                for (; s < 280; ++s) lengths[s] = 7;
                for (; s < 288; ++s) lengths[s] = 8;
                lencode.build(lengths, 288);
                for (s = 0; s < 30; ++s) lengths[s] = 5;
                distcode.build(lengths, 30);
            } else {
                static const short order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                int nlen = br.bits(5) + 257, ndist = br.bits(5) + 1, ncode = br.bits(4) + 4;
                for (int i = 0; i < 19; ++i) lengths[i] = 0;
                for (int i = 0; i < ncode; ++i) lengths[order[i]] = (short)br.bits(3);
                Huffman cl;
                cl.build(lengths, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = cl.decode(br);
                    if (sym < 16) lengths[idx++] = (short)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (!idx) throw std::runtime_error("png: bad lengths"); val = lengths[idx - 1]; rep = 3 + br.bits(2); }
                        else if (sym == 17) rep = 3 + br.bits(3);
                        else rep = 11 + br.bits(7);
                        if (idx + rep > nlen + ndist) throw std::runtime_error("png: bad lengths");
                        while (rep--) lengths[idx++] = (short)val;
                    }
                }
                lencode.build(lengths, nlen);
                distcode.build(lengths + nlen, ndist);
            }
            for (;;) {
                int sym = lencode.decode(br);
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) throw std::runtime_error("png: bad length symbol");
                    int len = lbase[sym] + br.bits(lext[sym]);
                    int ds = distcode.decode(br);
                    if (ds >= 30) throw std::runtime_error("png: bad distance symbol");
                    size_t dist = (size_t)dbase[ds] + (size_t)br.bits(dext[ds]);
                    if (dist > out.size()) throw std::runtime_error("png: distance too far back");
                    size_t from = out.size() - dist;
                    for (int i = 0; i < len; ++i) out.push_back(out[from + i]);
                }
            }
        } else {
            throw std::runtime_error("png: bad block type");
        }
    } while (!last);
}

int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

std::vector<uint8_t> read_file(const std::string& path)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) data.insert(data.end(), buf, buf + n);
    std::fclose(f);
    return data;
}

} // namespace

namespace imgio {

// stbi_write_png(path, w, h, 4, data, w*4) equivalent: 8-bit RGBA, filter 0, zlib "stored" blocks (valid, uncompressed).
void write_png_rgba8(const std::string& path, int w, int h, const uint32_t* rgba)
{
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * ((size_t)w * 4 + 1));
    for (int y = 0; y < h; ++y) {
        raw.push_back(0);
        const uint8_t* row = reinterpret_cast<const uint8_t*>(rgba + (size_t)y * w);
        raw.insert(raw.end(), row, row + (size_t)w * 4);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;
    size_t pos = 0;
    while (pos < raw.size() || raw.empty()) {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; ++i) { a = (a + raw[pos + i]) % 65521; b = (b + a) % 65521; }
        pos += n;
        if (raw.empty()) break;
    }
    put32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)w); put32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
}

// stbi_load(path, &w, &h, &comp, STBI_rgb_alpha) equivalent for non-interlaced 8-bit PNGs (grey, grey+alpha, RGB, RGBA, palette).
Image load_png_rgba8(const std::string& path)
{
    std::vector<uint8_t> d = read_file(path);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8)) throw std::runtime_error(path + ": not a PNG");
    size_t p = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    while (p + 8 <= d.size()) {
        const uint32_t len = ((uint32_t)d[p] << 24) | ((uint32_t)d[p + 1] << 16) | ((uint32_t)d[p + 2] << 8) | d[p + 3];
        std::string type(d.begin() + p + 4, d.begin() + p + 8);
        const uint8_t* body = d.data() + p + 8;
        if (p + 12 + len > d.size()) throw std::runtime_error(path + ": truncated chunk");
        if (type == "IHDR") {
            if (len < 13) throw std::runtime_error(path + ": short IHDR");
            const uint32_t uw = ((uint32_t)body[0] << 24) | ((uint32_t)body[1] << 16) | ((uint32_t)body[2] << 8) | body[3];
            const uint32_t uh = ((uint32_t)body[4] << 24) | ((uint32_t)body[5] << 16) | ((uint32_t)body[6] << 8) | body[7];
            if (uw == 0 || uh == 0 || uw > 65535u || uh > 65535u) throw std::runtime_error(path + ": image size out of range (1..65535 per side)");
            w = (int)uw; h = (int)uh;
            depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (type == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (type == "PLTE") plte.assign(body, body + len);
        else if (type == "tRNS") trns.assign(body, body + len);
        else if (type == "IEND") break;
        p += 12 + len;
    }
    if (w <= 0 || h <= 0 || depth != 8 || interlace != 0) throw std::runtime_error(path + ": only non-interlaced 8-bit PNGs are supported");
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) throw std::runtime_error(path + ": unsupported colour type");
    if (idat.size() < 6) throw std::runtime_error(path + ": no image data");
    std::vector<uint8_t> raw;
    inflate(idat.data() + 2, idat.size() - 6, raw);
    size_t stride = (size_t)w * ch;
    if (raw.size() < (stride + 1) * (size_t)h) throw std::runtime_error(path + ": short image data");
    std::vector<uint8_t> px((size_t)h * stride);
    for (int y = 0; y < h; ++y) {
        int ft = raw[(size_t)y * (stride + 1)];
        const uint8_t* in = &raw[(size_t)y * (stride + 1) + 1];
        uint8_t* out = &px[(size_t)y * stride];
        const uint8_t* up = y ? out - stride : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= (size_t)ch ? out[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int v = in[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: throw std::runtime_error(path + ": bad filter");
            }
            out[i] = (uint8_t)v;
        }
    }
    Image img;
    img.width = w;
    img.height = h;
    img.rgba.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        uint32_t r, g, b, a = 255;
        const uint8_t* s = &px[i * ch];
        if (ctype == 0) { r = g = b = s[0]; }
        else if (ctype == 4) { r = g = b = s[0]; a = s[1]; }
        else if (ctype == 2) { r = s[0]; g = s[1]; b = s[2]; }
        else if (ctype == 6) { r = s[0]; g = s[1]; b = s[2]; a = s[3]; }
        else {
            size_t k = s[0];
            if (k * 3 + 2 >= plte.size()) throw std::runtime_error(path + ": palette index out of range");
            r = plte[k * 3]; g = plte[k * 3 + 1]; b = plte[k * 3 + 2];
            if (k < trns.size()) a = trns[k];
        }
        img.rgba[i] = r | (g << 8) | (b << 16) | (a << 24);
    }
    return img;
}

// stbi_load on a Radiance .hdr with STBI_rgb_alpha: RGBE decode, then stb's HDR->LDR conversion
// (extern/stb/stb_image.h:1864-1890: z = pow(x * h2l_scale_i, h2l_gamma_i) * 255 + 0.5, clamped, with h2l_gamma_i = 1/2.2,
// h2l_scale_i = 1, :1559), alpha 255.  The reference samples this 8-bit gamma-encoded image WITHOUT re-linearising it
// (utils/image_buffer.cpp:47-48, device.cu:31-39).
Image load_hdr_as_ldr_rgba8(const std::string& path)
{
    std::vector<uint8_t> d = read_file(path);
    size_t p = 0;
    auto line = [&]() {
        std::string s;
        while (p < d.size() && d[p] != '\n') s += (char)d[p++];
        ++p;
        return s;
    };
    std::string first = line();
    if (first != "#?RADIANCE" && first != "#?RGBE") throw std::runtime_error(path + ": not a Radiance HDR file");
    bool fmt = false;
    for (;;) {
        std::string s = line();
        if (s.empty()) break;
        if (s == "FORMAT=32-bit_rle_rgbe") fmt = true;
        if (p >= d.size()) throw std::runtime_error(path + ": truncated header");
    }
    if (!fmt) throw std::runtime_error(path + ": unsupported HDR format");
    std::string res = line();
    int h = 0, w = 0;
    if (std::sscanf(res.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) throw std::runtime_error(path + ": unsupported HDR orientation");
    if (w > 65535 || h > 65535) throw std::runtime_error(path + ": image size out of range (1..65535 per side)");
    // before anything is allocated: every scanline takes at least 4 bytes of the file (an RLE scanline header; a flat one 4 * w)
    if (p > d.size() || (size_t)h * 4 > d.size() - p) throw std::runtime_error(path + ": truncated pixel data");
    std::vector<uint8_t> rgbe((size_t)w * h * 4);
    for (int y = 0; y < h; ++y) {
        uint8_t* row = &rgbe[(size_t)y * w * 4];
        bool rle = w >= 8 && w < 32768 && p + 4 <= d.size() && d[p] == 2 && d[p + 1] == 2 && !(d[p + 2] & 0x80) && ((d[p + 2] << 8) | d[p + 3]) == w;
        if (!rle) { // flat scanline
            if (p + (size_t)w * 4 > d.size()) throw std::runtime_error(path + ": truncated pixel data");
            std::memcpy(row, &d[p], (size_t)w * 4);
            p += (size_t)w * 4;
            continue;
        }
        p += 4;
        for (int c = 0; c < 4; ++c) {
            int x = 0;
            while (x < w) {
                if (p >= d.size()) throw std::runtime_error(path + ": truncated RLE data");
                int count = d[p++];
                if (count > 128) {
                    count -= 128;
                    if (p >= d.size() || x + count > w) throw std::runtime_error(path + ": corrupt RLE data");
                    uint8_t v = d[p++];
                    while (count--) row[(x++) * 4 + c] = v;
                } else {
                    if (p + count > d.size() || x + count > w || count == 0) throw std::runtime_error(path + ": corrupt RLE data");
                    while (count--) row[(x++) * 4 + c] = d[p++];
                }
            }
        }
    }
    Image img;
    img.width = w;
    img.height = h;
    img.rgba.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t* q = &rgbe[i * 4];
        float f[3] = {0, 0, 0};
        if (q[3] != 0) {
            float f1 = (float)std::ldexp(1.0f, q[3] - (int)(128 + 8));
            f[0] = q[0] * f1; f[1] = q[1] * f1; f[2] = q[2] * f1;
        }
        uint32_t o[3];
        for (int k = 0; k < 3; ++k) {
            float z = (float)std::pow(f[k] * 1.0f, 1.0f / 2.2f) * 255 + 0.5f;
            if (z < 0) z = 0;
            if (z > 255) z = 255;
            o[k] = (uint32_t)(int)z;
        }
        img.rgba[i] = o[0] | (o[1] << 8) | (o[2] << 16) | (255u << 24);
    }
    return img;
}

// Vertical flip applied by the reference after every stbi_load (image_buffer.cpp:50-55, application.cpp:229-234).
void flip_vertical(Image& img)
{
    for (int y = 0; y < img.height / 2; ++y)
        for (int x = 0; x < img.width; ++x) std::swap(img.rgba[(size_t)y * img.width + x], img.rgba[(size_t)(img.height - 1 - y) * img.width + x]);
}

} // namespace imgio
