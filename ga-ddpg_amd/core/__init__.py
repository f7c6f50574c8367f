"""Host-side mirror of the reference's ``core`` package for the update-step path."""
