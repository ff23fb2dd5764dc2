"""ORACLE package: CPU restatement of the reference hot path.  Test
infrastructure only (see oracle/ref_ops.py header); never imported by the
product package ``gfs-3dseg_gws_amd``."""
