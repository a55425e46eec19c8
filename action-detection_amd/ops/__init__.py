"""Mirror of the reference's ``ops`` package for the hot path (only ``ssn_ops`` is in scope)."""
