defmodule NxSignalAMD.MixProject do
  use Mix.Project

  # Elixir host side of the MI355X-native STFT / iSTFT / FIR path.  NOT BUILT IN THE BUILD IMAGE (no BEAM):
  # see INTEGRATION.md.  `make -C ../nif` produces priv/nxsig_nif.so next to this project.
  def project do
    [
      app: :nx_signal_amd,
      version: "0.1.0",
      elixir: "~> 1.15",
      deps: [{:nx, "~> 0.11"}]
    ]
  end

  def application, do: [extra_applications: [:logger]]
end
