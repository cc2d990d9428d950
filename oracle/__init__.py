"""CPU oracle for the fused MVulD hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Plain PyTorch fp32 (CPU) restatements of the reference's arithmetic for the
path BASELINE.json's north_star names:

  * ``swin_ref``     SwinV2 forward_features   (mvuld/models/swin_transformer_v2.py)
  * ``roberta_ref``  UniXcoder encoder + masked mean pool (mvuld/models/unixcoder.py:33-38
                     over HF transformers==4.18.0 RobertaModel -- third party)
  * ``gat_ref``      DGL 0.8.1 GATConv (third party, restated from its documented algorithm)
  * ``head_ref``     Multi_DefectModel_new_GCN.forward (mvuld/models/GraphModel.py:150-211)
                     + Rs_GCN (mvuld/models/Rs_GCN.py:52-73)
  * ``fused_ref``    composition of the three + CrossEntropy (main_bigvul.py:328-333)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / the reported CPU
baseline.  Nothing under ``mvuld_amd/`` imports it; the product path fails
loudly when ``libmvuld_hip.so`` is missing.

Pinning status (see DESIGN.md "Oracle"):
  * swin_ref, Rs_GCN and the head's own forward text are pinned against the
    reference modules themselves, imported in the build container by
    ``tests/golden/make_golden.py`` (outputs committed under tests/golden/).
  * GATConv (dgl-cu102==0.8.1) and RobertaModel (transformers==4.18.0) are
    third-party dependencies absent from /root/reference: their arithmetic is
    restated from the published algorithm; roberta_ref is additionally checked
    against the installed transformers 5.15.0 RobertaModel driven with the
    equivalent 4-D additive mask.  No reference test covers either =>
    **parity unpinned** for those two pieces.
"""
