"""Mirror of the reference's `lib` package for the hot path (same module names and signatures)."""
