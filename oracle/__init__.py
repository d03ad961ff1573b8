"""oracle/ -- TEST INFRASTRUCTURE.  CPU restatements of the reference's hot-path algorithm.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product (image-retrieval-wavelet_amd/) never does and fails loudly without its HIP library.
"""
