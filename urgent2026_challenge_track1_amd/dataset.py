"""Batch contract, data-parallel shard rule and audio I/O of the training loop.

Mirrors ``baseline_code/dataset.py``: ``read_kv_scp`` (:79-86), ``PreSimulatedDataset`` (:104-151),
``GroupedBatchSampler`` (:338-401: per-fs groups, length-sorted, ``indices[rank::world]`` shard at :361, buckets of
100*batch, shuffles seeded with ``random.seed(epoch + rank)``), ``collate_fn`` (:404-441: right-pad, batch =
``(clean[B,1,T], noisy[B,1,T], fs int32 0-d, lengths int32[B])``) and ``AudioDataModule`` (:444-524; validation is not
sharded).  ``soundfile`` is not available here, so WAV I/O is a small RIFF reader/writer (PCM16/24/32, float32).
``SyntheticPairDataset`` is the SURVEY 8(d) generator used by ``bench.py`` and the smoke test.
"""
import random
import struct
from collections import defaultdict

import numpy as np
import torch
from torch.utils.data import BatchSampler, DataLoader


def read_kv_scp(scp):
    rtv = {}
    with open(scp, "r") as f:
        for line in f:
            uid, value = line.strip().split()
            assert uid not in rtv, uid
            rtv[uid] = value
    return rtv


def read_audio(path):
    """-> (float32 [1, T], fs).  RIFF/WAVE PCM 16/24/32-bit or IEEE float32, first channel only kept as [1,T]."""
    with open(path, "rb") as f:
        data = f.read()
    assert data[:4] == b"RIFF" and data[8:12] == b"WAVE", "not a RIFF/WAVE file: %s" % path
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    tag, ch, fs, _, _, bits = fmt
    if tag == 0xFFFE:   # WAVE_FORMAT_EXTENSIBLE: sub-format in the first 2 bytes of the GUID
        tag = 3 if bits == 32 and b"\x03\x00" == data[data.find(b"fmt ") + 32:data.find(b"fmt ") + 34] else 1
    if tag == 3 and bits == 32:
        x = np.frombuffer(pcm, dtype="<f4").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm, dtype="<i4").astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
    else:
        raise ValueError("unsupported WAV encoding tag=%d bits=%d (%s)" % (tag, bits, path))
    x = x.reshape(-1, ch)[:, :1].T
    return np.ascontiguousarray(x), fs


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


class PreSimulatedDataset(torch.utils.data.Dataset):
    def __init__(self, clean_speech, noisy_speech, utt2fs, speech_length, max_duration=-1):
        self.clean_speech = read_kv_scp(clean_speech)
        self.noisy_speech = read_kv_scp(noisy_speech)
        self.utt2fs = {k: int(v) for k, v in read_kv_scp(utt2fs).items()}
        self.speech_length = {k: int(v) for k, v in read_kv_scp(speech_length).items()}
        self.uid = list(self.clean_speech.keys())
        self.max_duration = max_duration
        assert len(self.clean_speech) == len(self.noisy_speech) == len(self.utt2fs) == len(self.speech_length)

    def get_source_length(self):
        if self.max_duration > 0:
            return [min(self.speech_length[k], self.max_duration) for k in self.uid]
        return [self.speech_length[k] for k in self.uid]

    def get_srs(self):
        return [self.utt2fs[k] for k in self.uid]

    def __len__(self):
        return len(self.clean_speech)

    def __getitem__(self, index):
        uid = self.uid[index]
        audio, fs = read_audio(self.clean_speech[uid])
        assert fs == self.utt2fs[uid]
        noisy, fs = read_audio(self.noisy_speech[uid])
        assert fs == self.utt2fs[uid]
        if self.max_duration > 0 and audio.shape[1] > self.max_duration:   # max_duration is in SAMPLES (quirk C.4)
            start = random.randint(0, audio.shape[1] - self.max_duration)
            audio = audio[:, start:start + self.max_duration]
            noisy = noisy[:, start:start + self.max_duration]
        return audio, noisy, fs, audio.shape[1]


class SyntheticPairDataset(torch.utils.data.Dataset):
    """low-passed noise x 4 Hz envelope with 0.4 s near-silent edges (clean), + white noise at
    U(-5, 20) dB SNR (noisy), both peak-normalised to 0.9.  Deterministic per (seed, index)."""

    def __init__(self, n_items, fs_list=(48000,), seconds=4.0, seed=2024, vary_length=False):
        self.n, self.fs_list, self.seconds, self.seed, self.vary = n_items, list(fs_list), seconds, seed, vary_length

    def _len(self, i):
        fs = self.fs_list[i % len(self.fs_list)]
        L = int(self.seconds * fs)
        if self.vary:
            L = int(L * (0.6 + 0.4 * ((i * 2654435761) % 1000) / 1000.0))
        return fs, L

    def get_source_length(self):
        return [self._len(i)[1] for i in range(self.n)]

    def get_srs(self):
        return [self._len(i)[0] for i in range(self.n)]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        fs, L = self._len(i)
        rng = np.random.default_rng(self.seed * 1000003 + i)
        n = rng.standard_normal(L)
        k = np.fft.rfftfreq(L)
        clean = np.fft.irfft(np.fft.rfft(n) / (1.0 - 0.95 * np.exp(-2j * np.pi * k)), n=L)
        clean /= clean.std()
        t = np.arange(L) / fs
        clean *= 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 2 * np.pi))
        edge = min(int(0.4 * fs), L // 4)
        clean[:edge] *= 1e-3
        clean[L - edge:] *= 1e-3
        clean *= 0.9 / np.abs(clean).max()
        snr = rng.uniform(-5.0, 20.0)
        noise = rng.standard_normal(L)
        noise *= np.sqrt((clean ** 2).mean() / ((noise ** 2).mean() * 10 ** (snr / 10)))
        noisy = clean + noise
        sc = 0.9 / max(np.abs(noisy).max(), np.abs(clean).max())
        return (clean * sc).astype(np.float32)[None], (noisy * sc).astype(np.float32)[None], fs, L


class GroupedBatchSampler(BatchSampler):
    def __init__(self, dataset, batch_size, rank, world_size, seed=0, drop_last=False, bucket_size_mult=100,
                 sampler=None):
        self.batch_size = batch_size
        self.drop_last = drop_last
        self.bucket_size = batch_size * bucket_size_mult
        self.epoch = 0
        self.world_size = world_size
        self.rank = rank
        self.seed = seed
        self.generator = torch.Generator().manual_seed(seed + rank + self.epoch)
        sr_groups = defaultdict(list)
        for idx, sr in enumerate(dataset.get_srs()):
            sr_groups[sr].append(idx)
        self.buckets = []
        source_length = dataset.get_source_length()
        for sr, indices in sr_groups.items():
            sorted_indices = sorted(indices, key=lambda x: source_length[x])
            sorted_indices = sorted_indices[self.rank::self.world_size]      # the data-parallel shard rule
            for i in range(0, len(sorted_indices), self.bucket_size):
                self.buckets.append(sorted_indices[i:i + self.bucket_size])

    def set_epoch(self, epoch):
        self.epoch = epoch
        self.generator.manual_seed(self.seed + self.rank + self.epoch)

    def __iter__(self):
        random.seed(self.epoch + self.rank)
        random.shuffle(self.buckets)
        all_batches = []
        for bucket in self.buckets:
            random.shuffle(bucket)
            for i in range(0, len(bucket), self.batch_size):
                batch = bucket[i:i + self.batch_size]
                if len(batch) < self.batch_size and self.drop_last:
                    continue
                all_batches.append(batch)
        random.shuffle(all_batches)
        return iter(all_batches)

    def state_dict(self):
        return {"seed": self.seed, "epoch": self.epoch}

    def __len__(self):
        total = 0
        for bucket in self.buckets:
            n = len(bucket)
            total += n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size
        return total


def collate_fn(batch):
    speechs = [torch.as_tensor(item[0]) for item in batch]
    noisy_speechs = [torch.as_tensor(item[1]) for item in batch]
    srs = [item[2] for item in batch]
    lengths = [item[3] for item in batch]
    assert all(sr == srs[0] for sr in srs), "mixed sampling rates in one batch"
    max_length = max(a.shape[1] for a in speechs)
    pad = lambda a: torch.nn.functional.pad(a, (0, max_length - a.shape[1]), value=0.0)
    return (torch.stack([pad(a) for a in speechs], dim=0), torch.stack([pad(a) for a in noisy_speechs], dim=0),
            torch.tensor(srs[0], dtype=torch.int32), torch.tensor(lengths, dtype=torch.int32))


class AudioDataModule:
    def __init__(self, config, rank=0, world_size=1):
        self.config, self.rank, self.world_size = config, rank, world_size
        self.num_worker, self.batch_size = config.num_worker, config.batch_size
        td, vd = config.train_set_path, config.valid_set_path
        if str(td).startswith("synthetic"):            # "synthetic[:n_items]" -> on-the-fly SURVEY 8(d) pairs
            n = int(str(td).split(":")[1]) if ":" in str(td) else 64
            self.train_dataset = SyntheticPairDataset(n, seed=config.seed)
            self.val_dataset = SyntheticPairDataset(max(self.batch_size, n // 8), seed=config.seed + 1)
        else:
            if config.train_set_dynamic_mixing:
                raise NotImplementedError("DynamicMixingDataset (dataset.py:154-335) is the next §8(f) row")
            mk = lambda d, md: PreSimulatedDataset("%s/spk1.scp" % d, "%s/wav.scp" % d, "%s/utt2fs" % d,
                                                   "%s/speech_length.scp" % d, max_duration=md)
            self.train_dataset = mk(td, config.max_duration)
            self.val_dataset = mk(vd, -1)
        self.train_batch_sampler = None

    def _loader(self, ds, sampler):
        nw = self.num_worker
        return DataLoader(ds, batch_sampler=sampler, num_workers=nw, pin_memory=False,
                          persistent_workers=nw > 0, collate_fn=collate_fn)

    def train_dataloader(self):
        self.train_batch_sampler = GroupedBatchSampler(self.train_dataset, self.batch_size, self.rank,
                                                       self.world_size, drop_last=True)
        return self._loader(self.train_dataset, self.train_batch_sampler)

    def val_dataloader(self):       # not sharded: every rank validates the full set (dataset.py:507-516)
        return self._loader(self.val_dataset, GroupedBatchSampler(self.val_dataset, self.batch_size, 0, 1,
                                                                  drop_last=True))
