"""The legs of bench.py beside the headline, one module each: the contract line builder, the headline (`run_infer`), the CPU baseline and the rank
launch stay in bench.py, which re-exports what is here.  Shared constants and timing / accounting helpers: `benchlegs.common`."""
