/* Image buffer, pixel quantisation and BMP I/O of the host (reference:
 * src/image.c).  The hot path's pixel store runs on the GPU
 * (caster_hip_resolve); image_buffer_set_pixel is kept for host-side callers. */
#include "craytracer.h"

#include <stdlib.h>
#include <string.h>

img_buffer* new_image_buffer(u32 w, u32 h, u8 bpp)          /* src/image.c:6-14 */
{
    img_buffer* b = (img_buffer*)malloc(sizeof(img_buffer));
    if (!b) return NULL;
    b->w = w; b->h = h; b->bpp = bpp;
    b->data = (u8*)calloc((size_t)w * h * bpp, 1);
    if (!b->data) { free(b); return NULL; }
    return b;
}

void image_buffer_set_all(img_buffer* buffer, v3_f32 color) /* src/image.c:16-28 */
{
    if (!buffer) return;
    const size_t n = (size_t)buffer->w * buffer->h * 4;
    for (size_t k = 0; k < n; k += 4) {
        buffer->data[k] = (u8)(color.x * 255);
        buffer->data[k + 1] = (u8)(color.y * 255);
        buffer->data[k + 2] = (u8)(color.z * 255);
        buffer->data[k + 3] = 0x00;
    }
}

static f32 clampf(f32 x, f32 lo, f32 hi) { return x < lo ? lo : (x > hi ? hi : x); }

void image_buffer_set_pixel(img_buffer* buffer, i32 x, i32 y, v3_f32 color)  /* src/image.c:30-36 */
{
    u8* px = buffer->data + 4 * ((size_t)y * buffer->w + x);
    px[0] = (u8)(clampf(color.x, 0.0f, 0.999f) * 256);
    px[1] = (u8)(clampf(color.y, 0.0f, 0.999f) * 256);
    px[2] = (u8)(clampf(color.z, 0.0f, 0.999f) * 256);
    px[3] = 0x00;
}

void free_image_buffer(img_buffer* buffer)                  /* src/image.c:55-63 */
{
    if (!buffer) return;
    free(buffer->data);
    free(buffer);
}

/* File layout of src/image.c:98-196: 14-byte file header, 40-byte info header,
 * 24 bits per pixel, rows padded to 4 bytes and written from the buffer's last
 * row to its first, pixel bytes in buffer order (R,G,B land in the B,G,R slots). */
long crt_bmp_encode(const img_buffer* buffer, u8* out)
{
    const int w = (int)buffer->w, h = (int)buffer->h;
    const int row = w * 3;
    const int pad = (4 - row % 4) % 4;
    const int stride = row + pad;
    const int size = 54 + stride * h;
    memset(out, 0, 54);
    out[0] = 'B'; out[1] = 'M';
    for (int k = 0; k < 4; k++) out[2 + k] = (u8)(size >> (8 * k));
    out[10] = 54;
    out[14] = 40;
    for (int k = 0; k < 4; k++) { out[18 + k] = (u8)(w >> (8 * k)); out[22 + k] = (u8)(h >> (8 * k)); }
    out[26] = 1;
    out[28] = 24;
    u8* p = out + 54;
    for (int y = h - 1; y >= 0; y--) {
        const u8* src = buffer->data + 4 * (size_t)y * w;
        for (int x = 0; x < w; x++) { p[0] = src[4 * x]; p[1] = src[4 * x + 1]; p[2] = src[4 * x + 2]; p += 3; }
        memset(p, 0, (size_t)pad);
        p += pad;
    }
    return (long)(p - out);
}

void write_img_buffer_to_file(img_buffer* buffer, const char* filename)     /* src/image.c:190-196 */
{
    if (!buffer) return;
    const int row = (int)buffer->w * 3;
    const size_t bytes = 54 + (size_t)(row + (4 - row % 4) % 4) * buffer->h;
    u8* tmp = (u8*)malloc(bytes);
    if (!tmp) return;
    const long n = crt_bmp_encode(buffer, tmp);
    FILE* f = fopen(filename, "wb");
    if (f) { fwrite(tmp, 1, (size_t)n, f); fclose(f); }
    free(tmp);
}

static u32 rd32(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

img_buffer* read_bitmap_image(const char* filename)
{
    FILE* f = fopen(filename, "rb");
    if (!f) return NULL;
    u8 hdr[54];
    if (fread(hdr, 1, 54, f) != 54 || hdr[0] != 'B' || hdr[1] != 'M') { fclose(f); return NULL; }
    const u32 off = rd32(hdr + 10);
    const i32 w = (i32)rd32(hdr + 18);
    i32 h = (i32)rd32(hdr + 22);
    const u32 bits = (u32)hdr[28] | ((u32)hdr[29] << 8);
    const u32 compression = rd32(hdr + 30);
    const int top_down = h < 0;
    if (top_down) h = -h;
    if (w <= 0 || h <= 0 || (bits != 24 && bits != 32) || compression != 0) { fclose(f); return NULL; }
    const u8 bpp = (u8)(bits / 8);
    const size_t stride = ((size_t)w * bpp + 3) & ~(size_t)3;   /* bmWidthBytes */
    /* As the reference does (src/image.c:87-91), bpp = bmWidthBytes / bmWidth and
     * rows are copied as stored.  Widths whose rows need padding are repacked. */
    img_buffer* img = new_image_buffer((u32)w, (u32)h, bpp);
    if (!img) { fclose(f); return NULL; }
    u8* rowbuf = (u8*)malloc(stride);
    if (!rowbuf || fseek(f, (long)off, SEEK_SET) != 0) { free(rowbuf); free_image_buffer(img); fclose(f); return NULL; }
    for (i32 y = 0; y < h; y++) {
        if (fread(rowbuf, 1, stride, f) != stride) { free(rowbuf); free_image_buffer(img); fclose(f); return NULL; }
        const i32 dst = top_down ? (h - 1 - y) : y;
        memcpy(img->data + (size_t)dst * w * bpp, rowbuf, (size_t)w * bpp);
    }
    free(rowbuf);
    fclose(f);
    return img;
}

/* Smooth bands plus blocks: distinct texels in both directions so that a wrong
 * u/v mapping or texel index shows up. */
img_buffer* crt_make_test_texture(u32 w, u32 h)
{
    img_buffer* img = new_image_buffer(w, h, 3);
    if (!img) return NULL;
    for (u32 y = 0; y < h; y++) {
        for (u32 x = 0; x < w; x++) {
            u8* px = img->data + 3 * ((size_t)y * w + x);
            const int land = (((x * 8) / w + (y * 4) / h) & 1);
            px[0] = (u8)(land ? 40 + (x * 160) / w : 20);
            px[1] = (u8)(land ? 120 + (y * 100) / h : 60 + (x * 40) / w);
            px[2] = (u8)(land ? 30 : 140 + (y * 100) / h);
        }
    }
    return img;
}
