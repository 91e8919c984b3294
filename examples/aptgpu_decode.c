/*
 * aptgpu_decode.c — minimal C caller of the drop-in boundary (include/aptgpu.h):
 *
 *     aptgpu_decode in.wav out.pgm [contrast: telemetry|percent|minmax] [--no-sync]
 *
 * What `noaa-apt in.wav -o out.png` does up to the grayscale image (main.rs:91-110,
 * noaa_apt.rs:114-192), minus PNG encoding: load -> decode -> contrast limits -> 8-bit image,
 * written as a binary PGM.  Plain C99, links only libaptgpu.so.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aptgpu.h"

static void on_status(float progress, const char *text, void *user)
{
    (void)user;
    fprintf(stderr, "[%3.0f%%] %s\n", progress * 100.f, text);
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s in.wav out.pgm [telemetry|percent|minmax] [--no-sync]\n", argv[0]);
        return 2;
    }
    int contrast = APTGPU_CONTRAST_PERCENT, sync = 1;
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "telemetry")) contrast = APTGPU_CONTRAST_TELEMETRY;
        else if (!strcmp(argv[i], "percent")) contrast = APTGPU_CONTRAST_PERCENT;
        else if (!strcmp(argv[i], "minmax")) contrast = APTGPU_CONTRAST_MINMAX;
        else if (!strcmp(argv[i], "--no-sync")) sync = 0;
    }

    /* the file image */
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *bytes = malloc(n > 0 ? (size_t)n : 1);
    if (!bytes || fread(bytes, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "read error\n"); return 1; }
    fclose(f);

    /* the `standard` profile (default_settings.toml:108-116) */
    aptgpu_settings settings;
    memset(&settings, 0, sizeof settings);
    settings.work_rate = 12480;
    settings.resample_atten = 30.f;
    settings.resample_delta_freq = 1000.f;
    settings.resample_cutout = 4800.f;
    settings.demodulation_atten = 25.f;

    aptgpu_context ctx;
    memset(&ctx, 0, sizeof ctx);
    ctx.status = on_status;
    ctx.mode = APTGPU_MODE_STRICT;

    char err[1024] = "";
    float *rows = NULL;
    size_t n_rows_px = 0;
    aptgpu_stats stats;
    uint32_t rate = 0;
    int rc = aptgpu_decode_wav(&ctx, &settings, bytes, (size_t)n, sync, &rows, &n_rows_px, &stats, &rate, err,
                               sizeof err);
    free(bytes);
    if (rc != APTGPU_OK) { fprintf(stderr, "decode failed (%d): %s\n", rc, err); return 1; }
    fprintf(stderr, "%u Hz, %llu sync frames, %llu rows\n", rate, (unsigned long long)stats.n_sync,
            (unsigned long long)(n_rows_px / 2080));

    uint8_t *image = NULL;
    size_t n_px = 0;
    aptgpu_image_result info;
    rc = aptgpu_process_gray(&ctx, rows, n_rows_px, contrast, 0.98f, APTGPU_ROTATE_NO, &image, &n_px, &info, err,
                             sizeof err);
    aptgpu_free(rows);
    if (rc != APTGPU_OK) { fprintf(stderr, "image stage failed (%d): %s\n", rc, err); return 1; }

    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    fprintf(o, "P5\n2080 %u\n255\n", info.height);
    fwrite(image, 1, (size_t)info.height * 2080u, o);
    fclose(o);
    aptgpu_free(image);
    fprintf(stderr, "wrote %s: 2080 x %u, contrast limits %g .. %g\n", argv[2], info.height, info.low, info.high);
    return 0;
}
