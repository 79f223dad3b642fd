/* front_logs.c -- the LOGGING variant of the reference's parser for libedge264_hipfront.so.
 *
 * edge264.h's log_cb (edge264.h:36-41, 65) is served by a second compilation of src/edge264_headers.c: the reference's own build makes
 * edge264_headers_log.o from it with -DLOGS "-DADD_VARIANT(f)=f##_log" (Makefile:129-144, 329) and src/edge264.c:202-218 points the decoder's
 * parse_nal_unit[] table at the *_log functions when a callback is given -- without that object edge264_alloc refuses every log_cb
 * (src/edge264.c:219-221), which is what this library did until round 6: the reference's own src/edge264_check.c, which always passes one, could
 * not get a decoder.  Same sources, where they lie (the farm of symlinks of this directory's Makefile), the same four leaf files replaced by
 * our emitters; nothing here but the two definitions the reference's Makefile passes on the command line. */
#define LOGS
#define ADD_VARIANT(f) f##_log
#include E264_FARM_HEADERS_C
