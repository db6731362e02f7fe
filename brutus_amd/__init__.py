"""brutus_amd: the brutus per-star grid-likelihood path on MI355X (drop-in for the matching
modules of brutus 0.8.3: `fitting`, `pdf`, `utils`, `cluster`, `filters`)."""
__version__ = "0.8.3+mi355x.4"
