"""CPU oracle for the dense-correspondence training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline -- never as
the thing that is shipped or measured as the MI355X path.

Contents
--------
``resnet_dilated_oracle``  py3 / plain-torch restatement of ``Resnet34_8s`` & friends
                           (third-party code absent from /root/reference -- PARITY UNPINNED,
                           see the module header).
``loss_oracle``            py3 / plain-torch restatement of ``PixelwiseContrastiveLoss`` and
                           ``loss_composer`` (pinned against the reference's own source text
                           executed in the authoring container: tests/golden/loss_ref_*.npz).
``loss_numpy``             independent float64 numpy restatement of the loss (second opinion).
``synth``                  seeded synthetic inputs.
``step``                   one reference training step (training.py:325-346) on CPU.
"""
