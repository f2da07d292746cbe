"""Audio file I/O without ``soundfile`` (absent from the image): RIFF/WAVE (PCM 16/24/32, IEEE float32) read / write and
FLAC read (``flac.py``).  ``read_audio`` mirrors what the reference gets from ``soundfile.read(..., always_2d=True)``
followed by ``audio[:, :1].T`` (simulate_data_from_param.py:347-349): float samples in [-1, 1), first channel, shape [1, T].
"""
import struct

import numpy as np


def _wav_chunks(data, path):
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file: %s" % path)
    pos = 12
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        yield cid, data[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)


def _read_wav(data, path):
    fmt = pcm = None
    for cid, body in _wav_chunks(data, path):
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            pcm = body
    if fmt is None or pcm is None:
        raise ValueError("WAV file without fmt / data chunk: %s" % path)
    tag, ch, fs, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:          # WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real tag
        tag = struct.unpack("<H", fmt[24:26])[0]
    if tag == 3 and bits == 32:
        x = np.frombuffer(pcm, dtype="<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(pcm, dtype="<f8").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm, dtype="<i4").astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm[:len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
    else:
        raise ValueError("unsupported WAV encoding tag=%d bits=%d (%s)" % (tag, bits, path))
    n = x.size // ch * ch
    return x[:n].reshape(-1, ch), fs


def read_audio(path):
    """-> (float32 [1, T], fs): first channel of a WAV or FLAC file."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] == b"fLaC":
        from .flac import decode_flac
        x, fs = decode_flac(data, path)
    else:
        x, fs = _read_wav(data, path)
    return np.ascontiguousarray(x[:, :1].T), fs


def audio_frames(path):
    """number of sample frames without decoding the payload where the container says so (``SoundFile.frames``)."""
    with open(path, "rb") as f:
        head = f.read(1 << 16)
    if head[:4] == b"fLaC":
        from .flac import flac_streaminfo
        info = flac_streaminfo(head, path)
        if info["total_samples"]:
            return info["total_samples"]
        return read_audio(path)[0].shape[1]
    if head[:4] == b"RIFF":
        try:
            pos, ch, bits = 12, None, None
            while pos + 8 <= len(head):
                cid, size = head[pos:pos + 4], struct.unpack("<I", head[pos + 4:pos + 8])[0]
                if cid == b"fmt ":
                    _, ch, _, _, _, bits = struct.unpack("<HHIIHH", head[pos + 8:pos + 24])
                elif cid == b"data" and ch:
                    return size // (ch * bits // 8)
                pos += 8 + size + (size & 1)
        except struct.error:
            pass
    return read_audio(path)[0].shape[1]


def write_audio(path, x, fs, subtype="PCM_16"):
    x = np.asarray(x, dtype=np.float32).reshape(-1)
    if subtype == "FLOAT":
        pcm, tag, bits = x.astype("<f4").tobytes(), 3, 32
    else:
        pcm, tag, bits = np.clip(np.round(x * 32768.0), -32768, 32767).astype("<i2").tobytes(), 1, 16
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(pcm), b"WAVE", b"fmt ", 16, tag, 1, fs, fs * bits // 8,
                      bits // 8, bits, b"data", len(pcm))
    with open(path, "wb") as f:
        f.write(hdr + pcm)
