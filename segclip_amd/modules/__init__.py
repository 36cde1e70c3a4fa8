"""Host-side mirror of the reference's `modules` package for the hot path: same class names,
constructor / forward signatures and state_dict keys."""
