/*
 * oracle/_ref tool (test infrastructure): decodes a JPEG with the stb_image.h that the
 * reference vendors (external/stb_image.h, used by img_loader.h:38-44), compiled from
 * where it lies under /root/reference -- no reference source is copied into this repo.
 * Output: binary PPM (P6) of the 3-channel decode, exactly the bytes image_texture uploads
 * (textures.cuh:89-127).  Usage: stb_decode in.jpg out.ppm
 */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_FAILURE_USERMSG
#include MORT_STB_IMAGE_PATH
#include <stdio.h>

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s in.jpg out.ppm\n", argv[0]); return 2; }
    int w, h, n;
    unsigned char *d = stbi_load(argv[1], &w, &h, &n, 3);
    if (!d) { fprintf(stderr, "decode failed: %s\n", stbi_failure_reason()); return 1; }
    FILE *f = fopen(argv[2], "wb");
    if (!f) return 1;
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    fwrite(d, 1, (size_t)w * h * 3, f);
    fclose(f);
    return 0;
}
