"""CPU oracle for the URGENT-2026 track-1 hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` legs of
``bench.py`` may import anything from this package.  The product path
(``urgent2026_challenge_track1_amd``) never does: it fails loudly when the HIP
library is missing.

Pinning status.  The reference delegates most arithmetic to un-vendored third-party packages
(espnet==202412, pesq, pystoi, fast_bss_eval, librosa / resampy / soxr) that are absent here and holds no tests or golden
vectors; what it DOES hold in-tree is run here and compared bit for bit, the vectors are committed under tests/golden/:

* ``bsrnn_ref``  - ``BandSplit`` (481-bin table, four rates) and the dual-path loop: PINNED bit-equal to the reference's in-tree
  twin ``baseline_code/models/bsrnn_flowse.py:16-86,288-307`` (``tests/golden/make_golden_bsrnn.py`` -> ``ref_bsrnn.npz``);
  architecture pinned by the parameter counts of ``conf/models/BSRNN_baseline.yaml:30-32``; the espnet ``MaskDecoder`` head and
  the ``m*x + r`` tail: restated, unpinned.
* ``flow_ref``   - PINNED bit-equal to the reference's own ``bsrnn_flowse.py``, ``odes.py``, ``sampling/`` (``make_golden_flow.py``).
* ``mix_ref``    - the numpy / scipy DSP subset of ``simulation/simulate_data_from_param.py``: PINNED to the reference's own
  functions and whole ``process_one_sample(on_the_fly=True)`` samples (``make_golden_mix.py``); espnet2's ``detect_non_silence``
  and the resampy resamplers (``resampy_resample``: package absent, restated from its documented filter parameters): unpinned.
* data path (sampler, collate, recipe draw, config) - PINNED to the reference's own classes (``ref_mix.npz``, ``ref_config.npz``).
* ``metrics_ref`` - pystoi 0.4.1 ESTOI and fast_bss_eval SDR restated from the published algorithms, cross-checked against scipy;
  the soxr-HQ-SPECIFICATION resampling filter (its specification is asserted, not libsoxr's bits): unpinned (packages absent).
* ``pesq_ref`` / ``pesq_tables`` - ITU-T P.862 / P.862.1 / P.862.2 restated; the 8 kHz Bark tables pinned by their own redundancy,
  bands 41-48 of the 16 kHz table reconstructed (measured sensitivity 0.05 MOS); float64 and float32-buffer variants make the
  same integer decisions; unpinned against the pesq package.
* ``stft_ref``, ``losses_ref`` - espnet2 Stft / MultiResL1SpecLoss / SISNRLoss restated on stock torch, cross-checked against an
  independent float64 numpy DFT / manual formulas: unpinned.
"""
