// image_io.h -- the image readers of the dataset harnesses (no OpenCV: zlib only).
// PNG: 8/16-bit gray, gray+alpha, RGB, RGBA, non-interlaced; PGM: binary P5.  Colour is reduced to gray with
// cv::cvtColor's fixed-point weights; 16-bit samples are kept (depth maps) and also narrowed to their high byte.
#pragma once
#include <zlib.h>

#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace lvt_io {

struct Gray {
    int w = 0, h = 0;
    std::vector<unsigned char> px;   // 8-bit gray (16-bit files: the high byte)
    std::vector<uint16_t> px16;      // filled for 16-bit gray files only (TUM depth maps)
};

inline bool read_file(const std::string &path, std::vector<unsigned char> &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0);
    out.resize((size_t)n);
    f.read(reinterpret_cast<char *>(out.data()), n);
    return (bool)f;
}

inline uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// cv::cvtColor(BGR2GRAY) on 8-bit data: (R*4899 + G*9617 + B*1868 + 8192) >> 14
inline unsigned char to_gray(int r, int g, int b) { return (unsigned char)((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14); }

inline bool decode_png(const std::vector<unsigned char> &buf, Gray &img, std::string &err) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (buf.size() < 33 || std::memcmp(buf.data(), sig, 8) != 0) return err = "not a PNG", false;
    size_t off = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat;
    while (off + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[off]);
        const char *type = reinterpret_cast<const char *>(&buf[off + 4]);
        if (off + 12 + len > buf.size()) return err = "truncated chunk", false;
        const unsigned char *data = &buf[off + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            w = (int)be32(data), h = (int)be32(data + 4);
            depth = data[8], ctype = data[9], interlace = data[12];
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4))
            break;
        off += 12 + len;
    }
    int ch = 0;
    switch (ctype) {
        case 0: ch = 1; break;
        case 2: ch = 3; break;
        case 4: ch = 2; break;
        case 6: ch = 4; break;
        default: return err = "palette PNGs are not supported", false;
    }
    if (w <= 0 || h <= 0 || (depth != 8 && depth != 16) || interlace) return err = "unsupported PNG layout (need 8/16-bit, non-interlaced)", false;
    const int bps = depth / 8, bpp = ch * bps;
    const size_t stride = (size_t)w * bpp;
    std::vector<unsigned char> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return err = "zlib inflate failed", false;
    std::vector<unsigned char> cur(stride), prev(stride, 0);
    img.w = w, img.h = h;
    img.px.resize((size_t)w * h);
    img.px16.clear();
    if (depth == 16 && ch <= 2) img.px16.resize((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const unsigned char *line = &raw[(stride + 1) * (size_t)y];
        const int ft = line[0];
        for (size_t i = 0; i < stride; i++) {
            const int a = (i >= (size_t)bpp) ? cur[i - bpp] : 0, b = prev[i], c = (i >= (size_t)bpp) ? prev[i - bpp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return err = "bad PNG filter", false;
            }
            cur[i] = (unsigned char)(line[1 + i] + pred);
        }
        for (int x = 0; x < w; x++) {
            const unsigned char *p = &cur[(size_t)x * bpp];
            // 16-bit samples: the most significant byte (what an 8-bit imread returns)
            if (ch <= 2) {
                img.px[(size_t)y * w + x] = p[0];
                if (!img.px16.empty()) img.px16[(size_t)y * w + x] = (uint16_t)((p[0] << 8) | p[1]);  // PNG samples are big-endian
            } else
                img.px[(size_t)y * w + x] = to_gray(p[0], p[bps], p[2 * bps]);
        }
        std::swap(cur, prev);
    }
    return true;
}

inline bool decode_pgm(const std::vector<unsigned char> &buf, Gray &img, std::string &err) {
    size_t pos = 0;
    auto token = [&]() -> std::string {
        for (;;) {
            while (pos < buf.size() && std::isspace(buf[pos])) pos++;
            if (pos < buf.size() && buf[pos] == '#') {
                while (pos < buf.size() && buf[pos] != '\n') pos++;
                continue;
            }
            break;
        }
        std::string t;
        while (pos < buf.size() && !std::isspace(buf[pos])) t.push_back((char)buf[pos++]);
        return t;
    };
    if (token() != "P5") return err = "not a binary PGM", false;
    const int w = std::atoi(token().c_str()), h = std::atoi(token().c_str()), mx = std::atoi(token().c_str());
    pos++;  // single whitespace after maxval
    if (w <= 0 || h <= 0 || mx <= 0 || mx > 255 || pos + (size_t)w * h > buf.size()) return err = "unsupported PGM", false;
    img.w = w, img.h = h;
    img.px.assign(buf.begin() + (long)pos, buf.begin() + (long)(pos + (size_t)w * h));
    return true;
}

// `stem` without extension: tries .png then .pgm
inline bool load_gray(const std::string &stem, Gray &img, std::string &err) {
    std::vector<unsigned char> buf;
    if (read_file(stem + ".png", buf)) return decode_png(buf, img, err);
    if (read_file(stem + ".pgm", buf)) return decode_pgm(buf, img, err);
    err = "no such file";
    return false;
}
// full path with extension
inline bool load_image(const std::string &path, Gray &img, std::string &err) {
    std::vector<unsigned char> buf;
    if (!read_file(path, buf)) return err = "no such file", false;
    if (path.size() > 4 && path.substr(path.size() - 4) == ".pgm") return decode_pgm(buf, img, err);
    return decode_png(buf, img, err);
}

}  // namespace lvt_io
