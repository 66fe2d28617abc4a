/* abi_example.c -- the C ABI used from plain C99: one pan/high-shelf source and one HRTF source through
 * gas_process_block, then the same HRTF source through the host layer.  Build (see tests/test_abi.py):
 *   gcc -std=c99 -Wall -Iinclude examples/abi_example.c -Lgodot-audio-spatializer_amd -lgas_amd -lm -o abi_example
 * Needs an MI355X to run; without one gas_ctx_create returns GAS_ERR_NO_DEVICE and the program says so. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gas_amd.h"
#include "gas_amd_host.h"

#define F 512

static int sine_stream(void *user, gas_audio_frame *buf, float rate, int frames) {
	int *pos = (int *)user;
	(void)rate;
	for (int i = 0; i < frames; i++) {
		float v = 0.25f * sinf(2.0f * 3.14159265f * 440.0f * (float)(*pos + i) / 48000.0f);
		buf[i].left = v;
		buf[i].right = v;
	}
	*pos += frames;
	return frames;
}

int main(void) {
	gas_config cfg;
	memset(&cfg, 0, sizeof cfg);
	cfg.struct_size = sizeof cfg;
	cfg.max_sources = 16;
	cfg.frames = F;
	cfg.channel_count = 1;
	cfg.mix_rate = 48000.0f;
	gas_ctx *ctx = NULL;
	int rc = gas_ctx_create(&cfg, &ctx);
	if (rc != GAS_OK) {
		printf("gas_ctx_create: %s\n", gas_strerror(rc));
		return rc == GAS_ERR_NO_DEVICE ? 0 : 1;
	}
	/* a synthetic 4-direction HRIR set: unit impulse left, delayed impulse right */
	float *hrir = (float *)calloc(4 * 2 * GAS_HRTF_TAPS, sizeof(float));
	for (int d = 0; d < 4; d++) {
		hrir[(d * 2 + 0) * GAS_HRTF_TAPS] = 1.0f;
		hrir[(d * 2 + 1) * GAS_HRTF_TAPS + 8 * d] = 0.5f;
	}
	rc = gas_hrtf_load(ctx, hrir, 4, GAS_HRTF_TAPS);
	uint32_t s_pan = 0, s_hrtf = 0;
	int32_t chain[1] = { GAS_FX_HRTF };
	rc |= gas_source_alloc(ctx, GAS_KIND_3D_MIX, NULL, 0, &s_pan);
	rc |= gas_source_alloc(ctx, GAS_KIND_EFFECT, chain, 1, &s_hrtf);
	gas_params p;
	memset(&p, 0, sizeof p);
	p.mix_volumes[0][0] = 0.8f;
	p.mix_volumes[0][1] = 0.3f;
	p.pitch_scale = 1.0f;
	p.linear_attenuation = 0.5f;
	p.attenuation_filter_cutoff_hz = 5000.0f;
	p.hrtf_gain = 1.0f;
	p.hrtf_dir = 2;
	rc |= gas_params_publish(ctx, s_pan, &p);
	rc |= gas_params_publish(ctx, s_hrtf, &p);
	static gas_audio_frame src[2][F], out[F];
	float peaks[2][2];
	int pos = 0;
	sine_stream(&pos, src[0], 1.0f, F);
	memcpy(src[1], src[0], sizeof src[0]);
	uint32_t slots[2];
	slots[0] = s_pan;
	slots[1] = s_hrtf;
	rc |= gas_process_block(ctx, &src[0][0], slots, 2, F, out, &peaks[0][0], GAS_MEM_HOST);
	printf("gas_process_block: %s; out[100] = (%f, %f); peaks pan (%f, %f) hrtf (%f, %f)\n", gas_strerror(rc), out[100].left, out[100].right, peaks[0][0], peaks[0][1], peaks[1][0], peaks[1][1]);

	gas_host *host = NULL;
	rc |= gas_host_create(ctx, GAS_KIND_EFFECT, chain, 1, &host);
	uint32_t id = 0;
	int pos2 = 0;
	rc |= gas_host_start_playback(host, sine_stream, &pos2, &id);
	rc |= gas_host_set_spatializer_parameters(host, id, &p);
	for (int cb = 0; cb < 3; cb++) {
		rc |= gas_host_get_mixed_frames(host, 0, out, F);
	}
	printf("gas_host_get_mixed_frames: %s; out[100] = (%f, %f); playbacks %d\n", gas_strerror(rc), out[100].left, out[100].right, gas_host_playback_count(host));
	gas_host_destroy(host);
	gas_ctx_destroy(ctx);
	free(hrir);
	return rc == GAS_OK ? 0 : 1;
}
