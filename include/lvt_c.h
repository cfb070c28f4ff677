/*
 * lvt_c.h -- C-ABI of the MI355X-native LVT tracking hot path (drop-in boundary).
 *
 * These five entry points are exactly what the reference's shared library `lvt_c` exports
 * (reference: lvt/src/lvt_c.h:55-62, implementation lvt/src/lvt_c.cpp:33-148).  A caller that
 * links the reference's liblvt_c.so can link this library instead with no source change:
 * same names, same argument order and types, same "silent failure" behaviour (no error codes;
 * outputs untouched on failure; LOST is sticky and returns the last pose).
 *
 * All images are BORROWED host buffers: tightly packed 8-bit row-major (stride == n_cols),
 * read-only, valid for the duration of the call only.  R is the row-major camera-to-world
 * rotation, t the camera position, both in the first frame's left-camera frame.
 *
 * Everything behind this boundary runs as hand-written gfx950 HIP kernels; there is no CPU
 * fallback.  If no HIP device is usable lvt_create() returns NULL.
 *
 * Capacities (compile-time, lvt_amd/csrc/lvt_dev.h; the reference's containers grow without
 * bound): 4096 features per image after BRIEF's border filter, 16384 external corners per
 * list, 32768 map points, 16384 staged points, 64 detection cells.  Exceeding one never passes
 * silently: the frame is tracked on what fits and lvt_amd_last_error() / the `overflow` count
 * (include/lvt_amd_ext.h) say which capacity was hit.  Every shipped configuration of the
 * reference (KITTI, EuRoC, TUM) stays an order of magnitude below them.
 */
#ifndef LVT_C_INTERFACE_H__
#define LVT_C_INTERFACE_H__

#if defined(LVT_EXPORT_FUNCTIONS)
#define LVT_API __attribute__((visibility("default")))
#else
#define LVT_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef void *lvt_handle;

/* replaces reference lvt_c.cpp:33-48.  sensor_type: 1 = STEREO, 2 = RGBD.  Reads every key of the
 * reference's YAML (lvt_parameters.cpp:54-93); a missing key reads as 0, exactly as there.
 * Returns NULL on unreadable file / bad sensor_type / no usable GPU. */
LVT_API lvt_handle lvt_create(const char *config_file_name, int sensor_type);

/* replaces reference lvt_c.cpp:50-62. */
LVT_API void lvt_destroy(lvt_handle vo_system);

/* replaces reference lvt_c.cpp:64-89 (-> lvt_system::track, lvt_system.cpp:157-207). */
LVT_API void lvt_track(lvt_handle vo_system, unsigned char *left_img, unsigned char *right_img,
                       int n_rows, int n_cols, double R[3][3], double t[3]);

/* replaces reference lvt_c.cpp:91-132 (-> lvt_system::track_with_external_corners,
 * lvt_system.cpp:209-250): detection is skipped, BRIEF is computed at the given corners
 * (doubles narrowed to float; the 28-px border filter still applies). */
LVT_API void lvt_track_with_external_corners(lvt_handle vo_system, unsigned char *left_img,
                                             unsigned char *right_img, int n_rows, int n_cols,
                                             double corners_left[][2], int n_corners_left,
                                             double corners_right[][2], int n_corners_right,
                                             double R[3][3], double t[3]);

/* replaces reference lvt_c.cpp:134-148.  1 = not initialised, 2 = tracking, 3 = lost, -1 = error. */
LVT_API int lvt_get_status(lvt_handle vo_system);

#ifdef __cplusplus
}
#endif

#endif /* LVT_C_INTERFACE_H__ */
