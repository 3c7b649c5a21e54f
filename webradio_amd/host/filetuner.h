/*
 * filetuner.h -- a Tuner that replays a recording in the RTL-SDR byte format (unsigned 8-bit
 * interleaved IQ, what rtl_sdr(1) writes and what RtlSdrTuner::dataReady receives,
 * io/rtlsdrtuner.cxx:86-117).  The reference has no file source; BASELINE
 * config 1 needs one.  Patterned on the reference's RandSource (io/randsource.cxx): a
 * SampleSource whose process() fills the block.
 *
 *   setSubdevice(path)   the file to play (before start(), like every subdevice)
 *   setLoop(true)        rewind at the end instead of failing process()
 *
 * Samples are converted with the reference's rule (u8 - 128) / 128 (rtlsdrtuner.cxx:106).
 * The raw bytes of the current block stay available (RawU8Block), so the tuner batch can
 * ship 2 bytes per frame to the GPU and convert there.
 */
#ifndef FILETUNER_H_
#define FILETUNER_H_

#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "tuner.h"

using namespace std;

/* implemented by sources that still hold the current block in the RTL-SDR byte format */
class RawU8Block {
public:
	virtual ~RawU8Block() {}
	/* bytes of the block most recently produced (2 per frame), or NULL */
	virtual const uint8_t* rawU8(size_t *frames) const = 0;
	/* how many buffers the source alternates between: the bytes handed out for one block stay untouched until the
	 * source starts producing the block `rawU8Buffers()` later.  With 2 or more the GPU runtime lets the source refill
	 * one buffer while the transfer out of the other is still in flight (wr_dev_wait_uploads_but), so the host runs a
	 * block ahead of the GPU; with 1 the source's run() waits for the block's transfer first. */
	virtual unsigned int rawU8Buffers() const { return 1; }
};

/* implemented by sources whose current block already lies in GPU memory -- a capture card that writes there by DMA, a
 * resident recording, a synthetic source: nothing crosses PCIe, every GPU consumer reads the block where it lies, and the
 * tuner batch STREAMS it (wr_tuner_set_streaming: a doorbell per block instead of a kernel launch; r06).  The block's memory
 * must stay untouched until the source has produced two further blocks (rotate three buffers or more). */
struct wr_dev;
class DeviceBlock {
public:
	virtual ~DeviceBlock() {}
	/* device address of the block most recently produced (interleaved float pairs), the device context it was allocated
	 * through and its frames -- or NULL */
	virtual const float* deviceBlock(wr_dev **dev, size_t *frames) const = 0;
};

class FileTuner : public Tuner, public RawU8Block
{
public:
	FileTuner(const string &name = "<undefined>");
	virtual ~FileTuner();

	void setLoop(bool loop) { _loop = loop; }
	bool loop() const { return _loop; }
	unsigned long framesPlayed() const { return _played; }

	const uint8_t* rawU8(size_t *frames) const;
	unsigned int rawU8Buffers() const { return 2; }

	static Tuner* factory(const string &name);

protected:
	bool init();
	void deinit();
	bool process(const vector<sample_t> &inBuffer, vector<sample_t> &outBuffer);

private:
	FILE*			_file;
	bool			_loop;
	unsigned long	_played;
	vector<uint8_t>	_raw[2];		/* alternating: the GPU may still be reading the block before this one */
	unsigned int	_cur;
	size_t			_rawFrames;
};

#endif /* FILETUNER_H_ */
