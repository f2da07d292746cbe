"""CPU oracle for the URGENT-2026 track-1 hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package.  The product path
(``urgent2026_challenge_track1_amd``) never does: it fails loudly when the HIP
library is missing.

Pinning status (see DESIGN.md "Oracle"): the reference delegates all arithmetic
to un-vendored third-party packages (espnet==202412, pesq, pystoi,
fast_bss_eval) that are absent here and holds no tests / golden vectors, so
every module states its own pin:

* ``mix_ref``    -- the numpy / scipy DSP subset of ``simulation/simulate_data_from_param.py`` (mix_noise,
  add_reverberation, filtfilt high-pass, clipping, packet_loss, peak normalisation): calls the same scipy / numpy routines
  as the reference; espnet2's ``detect_non_silence`` restated -> that function UNPINNED.
* ``metrics_ref`` -- pystoi 0.4.1 ESTOI and fast_bss_eval SDR restated from the published algorithms; cross-checked against
  scipy (``resample_poly``, ``solve_toeplitz``); UNPINNED (packages absent).
* ``flow_ref``  - pinned against the reference's own ``bsrnn_flowse.py``,
  ``odes.py`` and ``sampling/`` imported in the build container
  (``tests/golden/make_golden.py`` generated the committed vectors).
* ``bsrnn_ref`` - architecture pinned by the parameter counts printed in
  ``conf/models/BSRNN_baseline.yaml:30-31`` and, structurally, by the in-tree
  twin ``bsrnn_flowse.py`` (same dual-path loop / BandSplit); numerics are
  stock ``torch`` CPU ops.  espnet numerics themselves: parity unpinned.
* ``stft_ref``, ``losses_ref`` - restated from the published algorithms
  (cross-checked against an independent float64 numpy DFT / manual formulas);
  parity unpinned by the reference.  There is no PESQ oracle (not built).
"""
