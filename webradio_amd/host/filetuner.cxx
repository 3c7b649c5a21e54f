/*
 * filetuner.cxx -- see filetuner.h.  Host I/O only; the conversion rule is the one line of
 * the reference that matters here, io/rtlsdrtuner.cxx:106.
 */
#include "filetuner.h"

#include <stdlib.h>

#include "debug.h"

FileTuner::FileTuner(const string &name)
	: Tuner(name, "FileTuner"), _file(NULL), _loop(false), _played(0), _cur(0), _rawFrames(0)
{
	_name = "file";
	_manufacturer = "webradio_amd";
	_product = "RTL-SDR format recording";
}

FileTuner::~FileTuner()
{
	if (_file)
		fclose(_file);
}

Tuner* FileTuner::factory(const string &name)
{
	return new FileTuner(name);
}

bool FileTuner::init()
{
	if (inputChannels() != 2) {
		LOG_ERROR("Expect IQ recording (2 channels)\n");
		return false;
	}
	_file = fopen(subdevice().c_str(), "rb");
	if (!_file) {
		LOG_ERROR("Cannot open recording '%s'\n", subdevice().c_str());
		return false;
	}
	_played = 0;
	_rawFrames = 0;
	return true;
}

void FileTuner::deinit()
{
	if (_file)
		fclose(_file);
	_file = NULL;
	vector<uint8_t>().swap(_raw[0]);
	vector<uint8_t>().swap(_raw[1]);
	_rawFrames = 0;
}

bool FileTuner::process(const vector<sample_t> &inBuffer, vector<sample_t> &outBuffer)
{
	const size_t want = outBuffer.size();           /* blockSize counts floats == bytes here */
	_cur ^= 1u;                                      /* the other buffer: the last block's bytes may still be on their way */
	vector<uint8_t> &raw = _raw[_cur];
	raw.resize(want);
	size_t got = 0;
	while (got < want) {
		size_t n = fread(raw.data() + got, 1, want - got, _file);
		got += n;
		if (got == want)
			break;
		if (!_loop || ftell(_file) == 0) {
			_rawFrames = 0;
			return false;                           /* end of the recording */
		}
		rewind(_file);
	}
	/* Every consumer takes the block from the device copy (staged from the BYTES and converted there with the
	 * same rule): nobody reads the float vector, and at 100 Msps filling it costs more host time than the whole
	 * GPU path.  WEBRADIO_NO_U8_STAGING=1 (the float block is what gets staged then) keeps it filled. */
	const char *nostage = getenv("WEBRADIO_NO_U8_STAGING");
	const bool skip = consumersReadOnDevice() && !(nostage && atoi(nostage));
	if (!skip)
		for (size_t n = 0; n < want; n++)
			outBuffer[n] = ((float)raw[n] - 128.0) / 128.0;   /* rtlsdrtuner.cxx:106 */
	setHostBlockValid(!skip);
	_rawFrames = want / 2;
	_played += _rawFrames;
	return true;
}

const uint8_t* FileTuner::rawU8(size_t *frames) const
{
	if (frames)
		*frames = _rawFrames;
	return _rawFrames ? _raw[_cur].data() : NULL;
}
