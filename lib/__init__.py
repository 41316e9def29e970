"""Drop-in module paths of the reference (`lib.preprocessing`, `lib.cython_impl.tools`,
`lib.proposed_architectures`): thin re-exports of sm_hpss_mtl_amd.lib."""
