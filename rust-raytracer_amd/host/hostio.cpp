// hostio -- C entry points over the headless host's image readers and writers (host/image_io.hpp), so that the
// Python harness and the tests drive the SAME RGBE reader and PNG writer the C++ host uses.  No device code.
#include <cstring>
#include <string>

#include "image_io.hpp"

static thread_local std::string g_hostio_error;

extern "C" {

const char *hostio_last_error(void) { return g_hostio_error.c_str(); }

// Reads a Radiance .hdr.  With out == NULL only the size is returned; otherwise out must hold width*height*3 floats.
int hostio_load_hdr(const char *path, uint32_t *width, uint32_t *height, float *out, size_t capacity_floats) {
    try {
        imageio::HdrImage img = imageio::load_hdr(path);
        if (width) *width = img.width;
        if (height) *height = img.height;
        if (out) {
            if (capacity_floats < img.rgb.size()) {
                g_hostio_error = "buffer too small";
                return -1;
            }
            std::memcpy(out, img.rgb.data(), img.rgb.size() * sizeof(float));
        }
        return 0;
    } catch (const std::exception &e) {
        g_hostio_error = e.what();
        return -1;
    }
}

int hostio_save_hdr(const char *path, uint32_t width, uint32_t height, const float *rgb, int rle) {
    try {
        imageio::save_hdr(path, width, height, rgb, rle != 0);
        return 0;
    } catch (const std::exception &e) {
        g_hostio_error = e.what();
        return -1;
    }
}

int hostio_save_png(const char *path, const uint32_t *argb, uint32_t width, uint32_t height) {
    try {
        imageio::save_png(path, argb, width, height);
        return 0;
    } catch (const std::exception &e) {
        g_hostio_error = e.what();
        return -1;
    }
}

}  // extern "C"
