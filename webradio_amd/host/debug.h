/*
 * debug.h -- logging macros of the host runtime.
 *
 * Same three names and the same "file(line): message" form on stderr as webradio's
 * src/debug.h:6-8 (radio.cxx:89 and the web handlers use them), so reference sources
 * compile against this tree unchanged.  WEBRADIO_QUIET=1 in the environment silences
 * the DEBUG level (the reference logs every buffer resize, dspblock.cxx:179-184).
 */
#ifndef WEBRADIO_AMD_DEBUG_H
#define WEBRADIO_AMD_DEBUG_H

#include <stdio.h>
#include <stdlib.h>

static inline int wr_log_debug_enabled(void)
{
	static int state = -1;
	if (state < 0) {
		const char *q = getenv("WEBRADIO_QUIET");
		state = (q && *q && *q != '0') ? 0 : 1;
	}
	return state;
}

#define WR_LOG_EMIT(fmt, ...)	fprintf(stderr, "%s(%d): " fmt, __FILE__, __LINE__, ##__VA_ARGS__)
#define LOG_DEBUG(fmt, ...)		do { if (wr_log_debug_enabled()) WR_LOG_EMIT(fmt, ##__VA_ARGS__); } while (0)
#define LOG_INFO(fmt, ...)		WR_LOG_EMIT(fmt, ##__VA_ARGS__)
#define LOG_ERROR(fmt, ...)		WR_LOG_EMIT(fmt, ##__VA_ARGS__)

#endif
