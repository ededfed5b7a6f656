"""Mirror of the reference package of the same name (see the modules inside)."""
