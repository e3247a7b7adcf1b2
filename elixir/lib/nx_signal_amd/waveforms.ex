defmodule NxSignalAMD.Waveforms do
  @moduledoc """
  The one waveform on the accelerated path's boundary: `NxSignal.Waveforms.sinc/1` (lib/nx_signal/waveforms.ex:451-457), the
  building block of `Filters.firwin/3`. Evaluated by the library's host numerics with the reference's f32 rounding rule
  (`sin(pi t) / (pi t)`, 1 at `t == 0`), bit-identical to the reference's values.
  """
  alias NxSignalAMD.NIF

  @doc "See `NxSignal.Waveforms.sinc/1`. Returns an f32 tensor of the input's shape (f64 for an f64 tensor)."
  def sinc(%Nx.Tensor{type: {:f, 64}} = t) do
    {:ok, out} = NIF.sinc_f64(Nx.to_binary(t)) |> NxSignalAMD.unwrap!()
    Nx.from_binary(out, :f64) |> Nx.reshape(Nx.shape(t))
  end

  def sinc(%Nx.Tensor{} = t) do
    shape = Nx.shape(t)
    {:ok, out} = NIF.sinc(t |> Nx.as_type(:f32) |> Nx.to_binary()) |> NxSignalAMD.unwrap!()
    Nx.from_binary(out, :f32) |> Nx.reshape(shape)
  end

  def sinc(number) when is_number(number), do: sinc(Nx.tensor(number, type: :f32))
end
